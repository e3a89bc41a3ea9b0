// pe_front.hpp -- the per-instance numeric path, written against a "team" (the threads that cooperate on one
// circuit instance).  On the GPU the team is one workgroup (pe_kernels.hip); tests/emu instantiates the same
// code with a one-thread team on the host purely to check indexing before a kernel is ever launched
// (test infrastructure: the product library never instantiates the host team).
//
// Reference semantics restated here (paths relative to the reference tree):
//   companion_update  circult::update_tr_step -> step_changed_tr_define of capacitor.h:106-128,
//                     inductor.h:134-160, PN_junction.h:440-476
//   eval_devices      iterate_{dc,tr,trop}_define of capacitor.h:132-155, inductor.h:164-195, VAC.h:162-179,
//                     PN_junction.h:358-402,478-503 (vlimit :58-109, limexp :10-16); IAC.h:148-160, the four generators,
//                     coupled_inductors.h:160-246, relay.h:84-95, nmosfet.h / pmosfet.h:91-122, BJT_NPN.h / BJT_PNP.h:122-146
//   stamp             MNA accumulation of circult::solve_once (circuit.h:1015-1110, mna.h:60-157)
//   factor/solve      replaces Eigen SparseLU compute()+solve() (circuit.h:1516-1518); the forward substitution is fused
//                     into the factorisation whenever a solve follows it (front_factor, `fuse`)
//   newton_converged  circult::solve convergence test (circuit.h:921-948)
#pragma once
#include "pe_device.hpp"

#include <cassert>
#include <cmath>

#if defined(__HIPCC__)
    #define PE_DEV __device__ __forceinline__
#else
    #define PE_DEV inline
#endif
// developer aid: -DPE_ASM_MARKS puts named comments into the device assembly (scripts/asm_regions.py counts instructions per region)
#if defined(PE_ASM_MARKS) && defined(__HIPCC__)
    #define PE_MARK(name) asm volatile("; PE_MARK " name ::: "memory")
#else
    #define PE_MARK(name) ((void)0)
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
        ST_NO_CONVERGENCE = -4,
        ST_INACCURATE = -6
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
            // coupled_inductors.h:160-198: Req = (2/dt) [[L1 M],[M L2]], Ueq = -v_prev - Req i_prev
            double const* par = V.cl_par + static_cast<long long>(b) * V.nCl * 3;
            for(int i = tm.tid(); i < V.nCl; i += tm.size())
            {
                double r11 = 0.0, r12 = 0.0, r22 = 0.0, u1 = 0.0, u2 = 0.0;
                if(dt > 0.0)
                {
                    double const L1 = par[3 * i], L2 = par[3 * i + 1], kc = par[3 * i + 2];
                    double const M = kc * sqrt(L1 * L2);
                    double const req_scale = 2.0 / dt;
                    r11 = req_scale * L1;
                    r12 = req_scale * M;
                    r22 = req_scale * L2;
                    double const v1 = volt(x, V.cl_n[4 * i]) - volt(x, V.cl_n[4 * i + 1]);
                    double const v2 = volt(x, V.cl_n[4 * i + 2]) - volt(x, V.cl_n[4 * i + 3]);
                    double const i1 = x[V.cl_k[2 * i]], i2 = x[V.cl_k[2 * i + 1]];
                    u1 = -v1 - (r11 * i1 + r12 * i2);
                    u2 = -v2 - (r12 * i1 + r22 * i2);
                }
                int const o = V.cl_dv[i];
                dv[o] = r11;
                dv[o + 1] = r12;
                dv[o + 2] = r22;
                dv[o + 3] = u1;
                dv[o + 4] = u2;
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
    // dynamic_only: a later Newton iteration of the same solve point -- the values that do not depend on x (companions, sources at
    // this t) are still in dv from the first one
    PE_DEV void eval_devices(Team const& tm, DevView const& V, int b, int mode, double t, double last_step, bool dynamic_only = false)
    {
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        bool const tr = mode == MODE_TR;
        if(!dynamic_only)
        {
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
            for(int i = tm.tid(); i < V.nCl; i += tm.size())
                for(int q = 0; q < 5; ++q) dv[V.cl_dv[i] + q] = 0.0;
        }
        {
            // time sources: TR at t, TROP / OP / DC at t = 0 (base.h:248-304 fallbacks; generators' iterate_dc = iterate_tr(0));
            // IAC has an empty iterate_dc (IAC.h:124-128)
            double const* par = V.ts_par + static_cast<long long>(b) * V.nTs * 8;
            for(int i = tm.tid(); i < V.nTs; i += tm.size())
            {
                double const* p = par + 8 * i;
                int const kind = V.ts_kind[i];
                double const tt = tr ? t : 0.0;
                double val = 0.0;
                if(kind == 0)
                {
                    if(tr || mode == MODE_TROP) val = p[0] * sin(p[1] * tt + p[2]);
                }
                else
                {
                    double const Vh = p[1], Vl = p[2], freq = p[3], duty = p[4], phase = p[5], trise = p[6], tfall = p[7];
                    double const T = 1.0 / freq;
                    double const t0 = tt + phase / (2.0 * 3.14159265358979323846) / freq;
                    double const tm_ = fmod(t0, T);
                    if(kind == 1) val = Vl + ((Vh - Vl) / T) * tm_;                       // sawtooth.h:94-98
                    else if(kind == 2)
                        val = tm_ < duty * T ? Vh : Vl;                                  // square.h:99-101
                    else if(kind == 3)                                                     // pulse.h:113-132
                    {
                        double const Ton = duty * T;
                        if(tm_ < trise) val = Vl + ((Vh - Vl) / fmax(trise, 1e-30)) * tm_;
                        else if(tm_ < Ton - tfall)
                            val = Vh;
                        else if(tm_ < Ton)
                            val = Vh - ((Vh - Vl) / fmax(tfall, 1e-30)) * (tm_ - (Ton - tfall));
                        else
                            val = Vl;
                    }
                    else                                                                   // triangle.h:94-103
                    {
                        double const amp = Vh - Vl;
                        if(tm_ < 0.5 * T) val = Vl + (2.0 * amp / T) * tm_;
                        else
                            val = Vh - (2.0 * amp / T) * (tm_ - 0.5 * T);
                    }
                }
                dv[V.ts_dv[i]] = val;
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
        }
        {
            // relays: the contact follows the coil voltage of the current iterate with hysteresis (relay.h:84-95)
            double const* par = V.rl_par + static_cast<long long>(b) * V.nRl * 2;
            int* eng = V.rl_engaged + static_cast<long long>(b) * V.nRl;
            for(int i = tm.tid(); i < V.nRl; i += tm.size())
            {
                double const vctrl = volt(x, V.rl_n[2 * i]) - volt(x, V.rl_n[2 * i + 1]);
                int e = eng[i];
                if(!e)
                {
                    if(vctrl >= par[2 * i]) e = 1;
                }
                else if(vctrl <= par[2 * i + 1])
                    e = 0;
                eng[i] = e;
                dv[V.rl_dv[i]] = e ? 0.0 : V.r_open;
            }
        }
        {
            // three-pin non-linear devices, re-linearised around the current iterate
            double const* par = V.n3_par + static_cast<long long>(b) * V.nN3 * 3;
            for(int i = tm.tid(); i < V.nN3; i += tm.size())
            {
                double const* p = par + 3 * i;
                int const kind = V.n3_kind[i], o = V.n3_dv[i];
                double const v0 = volt(x, V.n3_n[3 * i]), v1 = volt(x, V.n3_n[3 * i + 1]), v2 = volt(x, V.n3_n[3 * i + 2]);
                if(kind == 18 || kind == 19)
                {
                    // Shichman-Hodges level 1, pins D, G, S: nmosfet.h:91-122, pmosfet.h:91-120
                    double const Kp = p[0], lambda = p[1], Vth = p[2];
                    bool const nmos = kind == 18;
                    double const Vc = nmos ? v1 - v2 : v2 - v1;   // Vgs | Vsg
                    double const Vds = v0 - v2;
                    double const Vx = nmos ? Vds : -Vds;          // Vds | Vsd
                    double const Vov = Vc - Vth;
                    double Id = 0.0, gm = 0.0, gds = 0.0;
                    if(Vov <= 0.0) {}
                    else if(Vx < Vov)
                    {
                        double const B = Vov * Vx - 0.5 * Vx * Vx;
                        double const Ids = Kp * B * (1.0 + lambda * Vx);
                        double const dI = Kp * ((Vov - Vx) * (1.0 + lambda * Vx) + B * lambda);
                        Id = nmos ? Ids : -Ids;
                        gm = Kp * Vx * (1.0 + lambda * Vx);
                        gds = nmos ? dI : -dI;
                    }
                    else
                    {
                        double const Ids = 0.5 * Kp * Vov * Vov * (1.0 + lambda * Vx);
                        Id = nmos ? Ids : -Ids;
                        gm = Kp * Vov * (1.0 + lambda * Vx);
                        gds = nmos ? 0.5 * Kp * Vov * Vov * lambda : 0.5 * Kp * Vov * Vov * (-lambda);
                    }
                    dv[o] = gds;
                    dv[o + 1] = gm;
                    dv[o + 2] = Id - gm * Vc - gds * Vds;
                }
                else
                {
                    // forward-active Ebers-Moll, pins B, C, E: BJT_NPN.h:122-146, BJT_PNP.h:122-146 (plain exp, no limiting)
                    double const Is_eff = p[0], Ute = p[1], BetaF = p[2];
                    double const Vj = kind == 20 ? v0 - v2 : v2 - v0;   // Vbe | Veb
                    double const e = exp(Vj / Ute);
                    double const geq = Is_eff * e / Ute;
                    double const Ij = Is_eff * (e - 1.0);
                    double const gm = BetaF * geq;
                    dv[o] = geq;
                    dv[o + 1] = Ij - Vj * geq;
                    dv[o + 2] = gm;
                    dv[o + 3] = BetaF * Ij - gm * Vj;
                }
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
#ifndef PE_STAMP_UN
    #define PE_STAMP_UN 8  // slots per thread in flight in the stamp kernels of the split schedule (their own launch: registers to spare)
#endif
    // out[s] = sum over the contribution list of slot s, in list order; four slots per thread in flight (the three
    // dependent loads ptr -> src -> dv of one slot would otherwise be fully exposed)
    // (t0, T): this thread's index and the thread count of the group that shares the slot range [lo, hi)
    // `list` != null: the slots are list[lo .. n) instead of lo .. n
    // `out2` != null: the sum also goes to out2[perm[slot]] (the stamp kernel of the split schedule initialises the permuted work vector w)
    template <int UN = 4>
    PE_DEV void gather_contributions(int t0, int T, int const* ptr, int const* src, double const* dv, double* out, int lo, int n, int const* list = nullptr,
                                     double* out2 = nullptr, int const* perm = nullptr)
    {
        for(int base = lo + t0; base < n; base += UN * T)
        {
            int slot[UN];
            int e[UN], end[UN];
            double acc[UN];
#pragma unroll
            for(int q = 0; q < UN; ++q)
            {
                int const k = base + q * T;
                bool const ok = k < n;
                int const s = ok ? (list ? list[k] : k) : 0;
                slot[q] = s;
                e[q] = ok ? ptr[s] : 0;
                end[q] = ok ? ptr[s + 1] : 0;
                acc[q] = 0.0;
            }
            bool more = true;
            while(more)
            {
                more = false;
                int sr[UN];
#pragma unroll
                for(int q = 0; q < UN; ++q) sr[q] = e[q] < end[q] ? src[e[q]] : -1;
                double v[UN];
#pragma unroll
                for(int q = 0; q < UN; ++q) v[q] = dv[sr[q] >= 0 ? sr[q] >> 1 : 0];
#pragma unroll
                for(int q = 0; q < UN; ++q)
                    if(sr[q] >= 0)
                    {
                        acc[q] = (sr[q] & 1) ? acc[q] - v[q] : acc[q] + v[q];
                        ++e[q];
                        more = more || e[q] < end[q];
                    }
            }
#pragma unroll
            for(int q = 0; q < UN; ++q)
                if(base + q * T < n)
                {
                    out[slot[q]] = acc[q];
                    if(out2) out2[perm[slot[q]]] = acc[q];
                }
        }
    }

    template <class Team>
    PE_DEV void stamp(Team const& tm, DevView const& V, int b)
    {
        double const* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        gather_contributions(tm.tid(), tm.size(), V.a_ptr, V.a_src, dv, V.aval + static_cast<long long>(b) * V.nnzA, 0, V.nnzA);
        gather_contributions(tm.tid(), tm.size(), V.b_ptr, V.b_src, dv, V.rhs + static_cast<long long>(b) * V.rows, 0, V.rows);
    }
    // the same with the slots dealt out in CONTIGUOUS chunks to G groups of T threads (split schedule: one workgroup per chunk).
    // The matrix slots are in front-assembly order, so a chunk is a patch of the circuit and the device values it gathers are
    // few enough to stay in the CU's L1: a 64-byte sector of dv is fetched from L2 once per chunk instead of once per lane.
    // Newton iterations after the first of a solve point: only the slots an x-dependent device contributes to are gathered again
    // (same lists, same summation order: bit-identical to a full stamp); everything else still holds the first iteration's values.
    // with_w (split schedule): w = P rhs as well -- every gathered row writes its own entry of w, the rows the dynamic stamp does not
    // gather are copied from the rhs of the first iteration (no thread reads what another one writes: no barrier, one launch less)
    // rhs_full (first Newton iteration of a transient step whose matrix is known to hold the stamp of this dt, round 4): the matrix takes the
    // x-dependent slots only -- everything else in A (conductances, companion conductances 2C/dt, 2L/dt, incidence, g_min) is the same from one
    // time point to the next while dt and the parameters stay -- and the right-hand side (histories, sources at the new t) is gathered in full
    PE_DEV void stamp_dynamic_chunk(DevView const& V, int b, int g, int G, int t0, int T, bool with_w = false, bool rhs_full = false)
    {
        double const* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        double* w = with_w ? V.w + static_cast<long long>(b) * V.rows : nullptr;
        auto range = [&](int n, int& lo, int& hi)
        {
            int const c = (n + G - 1) / G;
            lo = g * c < n ? g * c : n;
            hi = lo + c < n ? lo + c : n;
        };
        int lo, hi;
        range(V.n_dyn_a, lo, hi);
        gather_contributions<PE_STAMP_UN>(t0, T, V.a_ptr, V.a_src, dv, V.aval + static_cast<long long>(b) * V.nnzA, lo, hi, V.dyn_a);
        if(rhs_full)
        {
            range(V.rows, lo, hi);
            gather_contributions<PE_STAMP_UN>(t0, T, V.b_ptr, V.b_src, dv, V.rhs + static_cast<long long>(b) * V.rows, lo, hi, nullptr, w, V.row_dst);
            return;
        }
        range(V.n_dyn_b, lo, hi);
        gather_contributions<PE_STAMP_UN>(t0, T, V.b_ptr, V.b_src, dv, V.rhs + static_cast<long long>(b) * V.rows, lo, hi, V.dyn_b, w, V.row_dst);
        if(with_w)
        {
            double const* rhs = V.rhs + static_cast<long long>(b) * V.rows;
            range(V.rows, lo, hi);
            for(int k = lo + t0; k < hi; k += T)
            {
                int const r = V.row_src[k];
                if(!V.row_dyn[r]) w[k] = rhs[r];
            }
        }
    }
    PE_DEV void stamp_chunk(DevView const& V, int b, int g, int G, int t0, int T, bool with_w = false)
    {
        double const* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        double* w = with_w ? V.w + static_cast<long long>(b) * V.rows : nullptr;
        auto range = [&](int n, int& lo, int& hi)
        {
            int const c = (n + G - 1) / G;
            lo = g * c < n ? g * c : n;
            hi = lo + c < n ? lo + c : n;
        };
        int lo, hi;
        range(V.nnzA, lo, hi);
        gather_contributions<PE_STAMP_UN>(t0, T, V.a_ptr, V.a_src, dv, V.aval + static_cast<long long>(b) * V.nnzA, lo, hi);
        range(V.rows, lo, hi);
        gather_contributions<PE_STAMP_UN>(t0, T, V.b_ptr, V.b_src, dv, V.rhs + static_cast<long long>(b) * V.rows, lo, hi, nullptr, w, V.row_dst);
    }

    // ================================================================================================
    // Multifrontal LU on the static assembly tree (pe_symbolic.cpp).
    //
    //   phase 1  WAVE fronts: every wavefront walks its own list of small subtrees (front order <= wave_m) with the
    //            dense front in a private LDS slot -- no workgroup barrier at all;
    //   phase 2  COOPERATIVE fronts: the whole workgroup, pivot panels (L: m x p, U: p x u) in LDS, per-pivot updates
    //            restricted to the panels, then ONE register-tiled rank-p update of the Schur block that streams the
    //            children's contributions from a global workspace and writes the update matrix once.
    //
    // Team interface used here: tid/size/sync/sync_or (workgroup), lanes() = wavefront width, wave_id()/n_waves(),
    // wave_fence() = make this wavefront's LDS/global writes visible to its own later reads.
    // ================================================================================================

    PE_DEV bool bad_pivot(double piv) { return piv == 0.0 || !(fabs(piv) <= 1.7976931348623157e308); }

    // exact idx / d for 0 <= idx < 2^22, 1 <= d <= 4096 (front-local index arithmetic), without an integer divide
    PE_DEV int fdiv(int idx, float rcp) { return static_cast<int>((static_cast<float>(idx) + 0.5f) * rcp); }

    // dst[0..n) = src[0..n)  (global -> LDS staging) with PE_UNROLL independent loads in flight per thread
    template <int UN>
    PE_DEV void stage_copy(double* dst, double const* src, int n, int t0, int T)
    {
        for(int base = t0; base < n; base += UN * T)
        {
            double r[UN];
#pragma unroll
            for(int q = 0; q < UN; ++q)
            {
                int const idx = base + q * T;
                r[q] = idx < n ? src[idx] : 0.0;
            }
#pragma unroll
            for(int q = 0; q < UN; ++q)
            {
                int const idx = base + q * T;
                if(idx < n) dst[idx] = r[q];
            }
        }
    }

    // ---- one front, executed by a TEAM: the whole workgroup (cooperative fronts) or a single wavefront (wave
    // fronts, through tm.wave_team()); Lp (m x p, ld m) and Up (p x u, ld p) live in the team's LDS region.
    //
    // Blocked right-looking LU of the pivot panels, block = 16 pivots:
    //   (a0) wavefront 0 factors the 16 x 16 diagonal block in LDS (no workgroup barrier inside),
    //   (a1) one thread per row below / per column right of the block solves against that block in registers,
    //   (b)  the trailing panels are updated by 16x16x4 fp64 MFMA tiles (tm.tile_*: v_mfma_f64_16x16x4_f64),
    // then the Schur block S = (children's contributions) - L21 * U12 is produced by MFMA tiles (K = p) that
    // PULL the children's contributions through the inverse maps f_inv and write S exactly once.
    // A run of single-front top levels handled by one workgroup (k_m2_factor_top_wide): a front whose Schur block STAYS in its LDS
    // image (V.f_keep) hands the image over to its parent, a chain link in mode 3 -- the link's front is that block, where it lies.
    struct ChainState
    {
        double* img{};  // the block (column major, leading dimension ld) ...
        int ld{};
        double* g{};    // ... and the update part of the right-hand-side column that belongs to it
    };

    // CHAIN (compile time): only the run kernel of the wide top levels carries the hand-over -- every other instantiation compiles to
    // the code it was before (the extra paths cost the 128-VGPR kernels registers: 17 spilled VGPRs in k_m2_factor_parts<4> when
    // they were decided at run time).
    template <class Team, bool CHAIN = false>
    PE_DEV bool front_factor(Team const& tm, DevView const& V, int b, int s_in, double* lds, int cap, int profile, bool fuse, ChainState* cs = nullptr)
    {
        int const s = tm.uniform(s_in);
        int const p = V.f_p[s], u = V.f_u[s], m = p + u;
        double const* a = V.aval + static_cast<long long>(b) * V.nnzA;
        double* arena = V.arena + static_cast<long long>(b) * V.arena_doubles;
        double* fac = V.factor + static_cast<long long>(b) * V.factor_doubles;
        // FULL mode (the whole m x m front fits the team's LDS region, column major, ld m): the children's update
        // matrices stream in once, unconditionally and coalesced, and land by LDS scatter -- one memory round trip per
        // child instead of index-dependent pulls.  PANEL mode (large fronts): only the pivot panels live in LDS and the
        // Schur block pulls the children's entries tile by tile through the inverse maps.
        int const ch0 = V.f_child_ptr[s], ch1 = V.f_child_ptr[s + 1];
        // FUSED forward substitution (`fuse`): the permuted right-hand side rides through the factorisation as one more
        // column g[m] of the front -- pivot part from w, update part from the children's update vectors -- so that after the
        // block loop g[0..p) is the forward-substituted solution of these pivots and g[p..m) this front's update vector.
        // Saves the separate forward pass over the factor panels whenever a factorisation is followed by a solve.
        int const mode = tm.uniform(V.f_mode[s]);  // fixed with the launch geometry (build_assembly_lists): same rule, one place
        // mode 3: a chain link whose front already sits in LDS -- the Schur block its child (the front before it in this run of top
        // levels, whole-front layout) left in place: no zeroing, no assembly of the child, no trip of that block through the arena.
        // Same values in the same cells, same block loop: bit-identical to the link assembled from the arena.
        bool const cont = CHAIN && mode == 3;
#if !defined(__HIPCC__)
        assert((!cont || (cs && cs->img)) && "a mode-3 front needs the image its child left in LDS");
#endif
        bool const full = mode == 0 || cont;
        bool const chain = mode == 2;  // the single child's update matrix IS this front: f_rel of the child is the identity
        bool const keep = CHAIN && cs && V.f_keep[s];  // this front's Schur block stays in the image for its parent (mode 3)
        int const ldl = cont ? cs->ld : pe_ld(m);  // LDS leading dimension of the L panel / of the whole image (odd: bank spread)
        int const ldu = full ? ldl : pe_ld(p);  // ... of the U panel
        int const nlds = full ? ldl * m : ldl * p + ldu * u;
#if !defined(__HIPCC__)
        assert((cont || nlds + (fuse ? m : 0) <= cap) && "front image + right-hand-side column overrun the team's LDS region");
#endif
        double* Lp = cont ? cs->img : lds;
        double* Up = Lp + ldl * p;
        double* g = cont ? cs->g : lds + nlds;  // [m] right-hand-side column (fuse)
        int const T = tm.size(), t0 = tm.tid();
        int const c0 = V.f_col0[s];
        double* w = V.w + static_cast<long long>(b) * V.rows;
        long long const ck0 = tm.clock();
        PE_MARK("prologue");
        // the entries of A owned by this front and its slice of the right-hand side: their loads depend on nothing a previous
        // front wrote -- requested first, they fly behind the zeroing of the front (and that front's stores still in flight)
        int const e0 = V.f_asm_ptr[s] + t0, e1 = V.f_asm_ptr[s + 1];
        int const pos0 = e0 < e1 ? V.asm_pos[e0] : 0;
        double const v0 = e0 < e1 ? a[V.asm_slot ? V.asm_slot[e0] : e0] : 0.0;
        double const w0 = (fuse && t0 < p) ? w[c0 + t0] : 0.0;
        PE_MARK("zero");
        if(cont)
        {
            // g[0..m) holds the child's update vector: the pivot part takes this front's slice of the right-hand side on top
            // (w + vc in the arena path, vc + w here: the same sum)
            if(fuse && t0 < p) g[t0] += w0;
            if(fuse)
                for(int i = t0 + T; i < p; i += T) g[i] += w[c0 + i];
        }
        else
        {
            for(int i = t0; i < nlds; i += T) lds[i] = 0.0;
            if(fuse)
            {
                if(t0 < m) g[t0] = w0;
                for(int i = t0 + T; i < m; i += T) g[i] = i < p ? w[c0 + i] : 0.0;
            }
            tm.sync_lds();
        }
        PE_MARK("place");
        auto place = [&](int pos, double v)
        {
            int const r = pos >> 16, c = pos & 0xffff;
            if(c < p) Lp[r + c * ldl] += v;
            else
                Up[r + (c - p) * ldu] += v;  // r < p: an entry of A owned by this front touches a pivot row or column
        };
        if(e0 < e1) place(pos0, v0);
        for(int e = e0 + T; e < e1; e += T) place(V.asm_pos[e], a[V.asm_slot ? V.asm_slot[e] : e]);
        long long const cka = tm.clock();
        PE_MARK("children");
        // full fence: the children's update matrices (global memory, written by other lanes / wavefronts) become visible
        if(ch1 > ch0 && !cont) tm.sync();
        else
            tm.sync_lds();
        if(cont) {}
        else if(chain)
        {
            // the child's update matrix IS this front (a long separator split into links): straight copies
            int const c = V.f_child[ch0];
            double const* Sc = arena + V.f_sptr[c];
            float const rm = 1.0f / static_cast<float>(m);
            for(int base = t0; base < m * p; base += 4 * T)
            {
                double v[4];
#pragma unroll
                for(int q = 0; q < 4; ++q) v[q] = Sc[base + q * T < m * p ? base + q * T : 0];
#pragma unroll
                for(int q = 0; q < 4; ++q)
                    if(base + q * T < m * p)
                    {
                        int const idx = base + q * T, cc = fdiv(idx, rm), r = idx - cc * m;
                        Lp[r + cc * ldl] += v[q];
                    }
            }
            float const rp = 1.0f / static_cast<float>(p);
            for(int base = t0; base < p * u; base += 4 * T)
            {
                double v[4];
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    int const idx = base + q * T;
                    int const ix = idx < p * u ? idx : 0;
                    int const cc = fdiv(ix, rp), r = ix - cc * p;
                    v[q] = Sc[r + (p + cc) * m];
                }
#pragma unroll
                for(int q = 0; q < 4; ++q)
                    if(base + q * T < p * u)
                    {
                        int const idx = base + q * T, cc = fdiv(idx, rp), r = idx - cc * p;
                        Up[r + cc * ldu] += v[q];
                    }
            }
            if(fuse)
            {
                double const* vc = Sc + m * m;  // the child's update vector sits behind its update matrix
                for(int i = t0; i < m; i += T) g[i] += vc[i];
            }
            tm.sync_lds();
        }
        else if(ch1 > ch0)
        {
            // Destination-centric assembly (pe_symbolic.hpp: build_assembly_lists): every LDS cell that receives anything from a
            // child is owned by ONE thread, which adds its sources in the children's order -- no barrier between children, and
            // the loads of all children (indices from the shared lists, values from this instance's arena) are in flight together.
            int const r_lo = V.gl_rptr[s], R = V.gl_rptr[s + 1] - r_lo;
            unsigned short const* dst = V.gl_dst + V.gl_ptr[s];
            int const* sr = V.gl_src + V.gl_sptr[s];
            // Rounds are taken two at a time (round r covers the first n_r cells; n_r falls quickly: most cells have one or two
            // sources).  Per sweep a thread keeps UN cells in flight and requests the indices of its NEXT batch before it waits
            // for the values of the current one, so a batch costs one memory latency, not two.
            constexpr int UN = 4;
            for(int r = 0; r < R; r += 2)
            {
                int const na = V.gl_cnt[r_lo + r], nb = r + 1 < R ? V.gl_cnt[r_lo + r + 1] : 0;
                int const *sa = sr, *sb = sr + na;
                sr = sb + nb;
                int d[UN], ia[UN], ib[UN];
                auto fetch_idx = [&](int base)
                {
#pragma unroll
                    for(int q = 0; q < UN; ++q)
                    {
                        int const c = base + q * T;
                        int const cc = c < na ? c : 0;
                        d[q] = dst[cc];
                        ia[q] = sa[cc];
                        ib[q] = nb > 0 ? sb[c < nb ? c : 0] : 0;  // (a lane past the round's prefix re-reads its first source and drops it)
                    }
                };
                if(t0 < na) fetch_idx(t0);
                for(int base = t0; base < na; base += UN * T)
                {
                    double va[UN], vb[UN];
                    int dd[UN];
#pragma unroll
                    for(int q = 0; q < UN; ++q)
                    {
                        va[q] = arena[ia[q]];
                        vb[q] = arena[ib[q]];
                        dd[q] = d[q];
                    }
                    if(base + UN * T < na) fetch_idx(base + UN * T);
#pragma unroll
                    for(int q = 0; q < UN; ++q)
                    {
                        int const c = base + q * T;
                        if(c < na && (fuse || dd[q] < nlds))
                        {
                            double acc = lds[dd[q]] + va[q];
                            if(c < nb) acc += vb[q];
                            lds[dd[q]] = acc;
                        }
                    }
                }
            }
            tm.sync_lds();
        }
        long long const ck1 = tm.clock();
        PE_MARK("blockloop");
        int const NW = tm.n_waves();
        constexpr int NB = 8;
        // A bad pivot is reported once per front, after the block loop: nothing below branches on or indexes by matrix values,
        // and the barriers inside the loop order LDS traffic only.
        int bad = 0;
        for(int k0 = 0; k0 < p; k0 += NB)
        {
            int const kb = p - k0 < NB ? p - k0 : NB;
            if(tm.single_wave())
            {
                // one wavefront owns the front (m <= 64): diagonal block, rows below it and columns right of it in ONE pass over
                // registers, cross-lane traffic on v_readlane (tm.block_step)
                tm.for_each_wave([&](int, int lane, int) { bad |= tm.block_step(Lp, ldl, m, Up, ldu, g, p, u, k0, kb, fuse, lane); });
            }
            else
            {
                // (a0) diagonal block, wavefront 0: LU of the kb x kb block in registers (tm.diag_lu8), L stored scaled
                tm.for_each_wave(
                    [&](int w, int lane, int NL)
                    {
                        if(w == 0) bad |= tm.diag_lu8(Lp + k0 + k0 * ldl, ldl, kb, lane);
                    });
                tm.sync_lds();
                // (a1) rows below the block (L) and columns right of it (U): one thread each (tm.panel_solve)
                tm.panel_solve(Lp, ldl, m, Up, ldu, g, p, u, k0, kb, fuse, t0, T);
            }
            tm.sync_lds();
            // (b) trailing update of both panels, one 16 x 16 tile per wavefront at a time
            {
                int const r0 = k0 + kb;
                if(fuse)  // g[r0..m) -= L[r0..m, block] * g[block]: every row of the front, pivot and update part alike
                    for(int r = r0 + t0; r < m; r += T)
                    {
                        double acc = g[r];
#pragma unroll
                        for(int kk = 0; kk < NB; ++kk)
                            if(kk < kb) acc -= Lp[r + (k0 + kk) * ldl] * g[k0 + kk];
                        g[r] = acc;
                    }
                int const trL = (m - r0 + 15) / 16, tcL = (p - r0 + 15) / 16;   // L panel: rows r0..m, cols r0..p
                int const trU = (p - r0 + 15) / 16, tcU = (u + 15) / 16;        // U panel: rows r0..p, all columns
                int const nL = trL * tcL, nU = trU * tcU;
                tm.for_each_wave(
                    [&](int w, int lane, int NL)
                    {
                        for(int tile = w; tile < nL + nU; tile += NW)
                        {
                            double* C;
                            double const* B;
                            int ldc, ldb, mr, nc, row0;
                            if(tile < nL)
                            {
                                int const tc = tile / trL, tr = tile - tc * trL;
                                row0 = r0 + 16 * tr;
                                int const col0 = r0 + 16 * tc;
                                C = Lp + row0 + col0 * ldl;
                                ldc = ldl;
                                B = Lp + k0 + col0 * ldl;
                                ldb = ldl;
                                mr = m - row0 < 16 ? m - row0 : 16;
                                nc = p - col0 < 16 ? p - col0 : 16;
                            }
                            else
                            {
                                int const q = tile - nL;
                                int const tc = q / trU, tr = q - tc * trU;
                                row0 = r0 + 16 * tr;
                                int const col0 = 16 * tc;
                                C = Up + row0 + col0 * ldu;
                                ldc = ldu;
                                B = Up + k0 + col0 * ldu;
                                ldb = ldu;
                                mr = p - row0 < 16 ? p - row0 : 16;
                                nc = u - col0 < 16 ? u - col0 : 16;
                            }
                            auto acc = tm.tile_load(C, ldc, mr, nc, lane);
                            tm.tile_mulsub(acc, Lp + row0 + k0 * ldl, ldl, B, ldb, mr, nc, kb, lane);
                            tm.tile_store(acc, C, ldc, mr, nc, lane);
                        }
                    });
            }
            tm.sync_lds();
        }
        if(tm.sync_or(bad)) return false;
        long long const ck2 = tm.clock();
        PE_MARK("schur");
        if(profile == 1 && V.prof && t0 == 0)
        {
            V.prof[b * PE_PROF + 6] += ck1 - ck0;
            V.prof[b * PE_PROF + 7] += ck2 - ck1;
        }
        // Schur block: S = (children's contributions) - L21 * U12
        if(u > 0)
        {
            double* Ss = arena + V.f_sptr[s];
            int const nt = (u + 15) / 16;
            // PANEL mode: the children's inverse maps (rows p.. of this front -> child index) are staged behind the panels
            // when they fit, so that a tile's pulls are ONE round of unconditional loads instead of index-dependent ones
            int const nch = ch1 - ch0;
            int const ng = fuse ? m + (m & 1) : 0;                      // the right-hand-side column sits right behind the panels
            long long* csp = reinterpret_cast<long long*>(lds + nlds + ng);  // per child: arena offset of its update matrix,
            int* cuc = reinterpret_cast<int*>(csp + nch);               //            its order,
            unsigned* cmk = reinterpret_cast<unsigned*>(cuc + nch);     //            the 16-row blocks of this front's update rows it touches,
            int* linv = cuc + 2 * nch;                                  //            its inverse map (rows p.. of this front)
            bool const staged = !full && !chain && nch <= 64 && (static_cast<long long>(nch) * (u + 4) + 2) * 4 <= static_cast<long long>(cap - nlds - ng) * 8;
            if(staged)
            {
                for(int q = t0; q < nch; q += T)
                {
                    int const cc = V.f_child[ch0 + q];
                    csp[q] = V.f_sptr[cc];
                    cuc[q] = V.f_u[cc];
                    cmk[q] = V.f_bmask[ch0 + q];
                }
                float const ru = 1.0f / static_cast<float>(u);
                for(int idx = t0; idx < nch * u; idx += T)
                {
                    int const c = fdiv(idx, ru), r = idx - c * u;
                    linv[idx] = V.f_inv[V.f_inv_off[ch0 + c] + p + r];
                }
                tm.sync_lds();
            }
            tm.for_each_wave(
                [&](int w, int lane, int NL)
                {
                    // chain link: the child's tile for the NEXT round is requested before this round's MFMA / store, so its
                    // (global, unconditional) loads fly behind the arithmetic
                    double const* Sch = chain ? arena + V.f_sptr[V.f_child[ch0]] : nullptr;
                    auto chain_tile = [&](int tile)
                    {
                        int const tj = tile / nt, ti = tile - tj * nt;
                        int const i0 = 16 * ti, j0 = 16 * tj;
                        int const mr = u - i0 < 16 ? u - i0 : 16, nc = u - j0 < 16 ? u - j0 : 16;
                        return tm.tile_load(Sch + (p + i0) + static_cast<long long>(p + j0) * m, m, mr, nc, lane);
                    };
                    auto nxt = tm.tile_zero();
                    if(chain && w < nt * nt) nxt = chain_tile(w);
                    if(staged)
                    {
                        // PANEL layout: software-pipelined over this wavefront's tiles -- the first two children of the NEXT tile are
                        // requested before the rank-p product of the current one, so a tile costs max(pull latency, product), not the sum.
                        // Only the children that touch a tile's row block AND column block are pulled (one ballot over the staged masks).
                        auto pull = [&](int q, int i0, int j0, int mr, int nc)
                        {
                            auto raw = tm.tile_zero();
                            int const uc = cuc[q];
                            double const* Sc = arena + csp[q];
                            int const* inv = linv + q * u;
                            tm.tile_foreach(raw, lane,
                                            [&](int r, int c, double& v)
                                            {
                                                bool const in = r < mr && c < nc;
                                                int const ci = in ? inv[i0 + r] : -1, cj = in ? inv[j0 + c] : -1;
                                                double const* src = (ci >= 0 && cj >= 0) ? Sc + (ci + cj * uc) : V.zero;
                                                v = *src;
                                            });
                            return raw;
                        };
                        auto first_two = [&](int tile, decltype(tm.tile_zero())& r0, decltype(tm.tile_zero())& r1)
                        {
                            int const tj = tile / nt, ti = tile - tj * nt;
                            int const i0 = 16 * ti, j0 = 16 * tj;
                            int const mr = u - i0 < 16 ? u - i0 : 16, nc = u - j0 < 16 ? u - j0 : 16;
                            auto todo = tm.tile_children(cmk, nch, ti, tj, lane);
                            r0 = tm.tile_zero();
                            r1 = tm.tile_zero();
                            if(todo)
                            {
                                r0 = pull(__builtin_ctzll(todo), i0, j0, mr, nc);
                                todo &= todo - 1;
                            }
                            if(todo)
                            {
                                r1 = pull(__builtin_ctzll(todo), i0, j0, mr, nc);
                                todo &= todo - 1;
                            }
                            return todo;  // children still to pull for this tile
                        };
                        auto n0 = tm.tile_zero(), n1 = tm.tile_zero();
                        decltype(tm.tile_children(cmk, nch, 0, 0, lane)) rest_next = 0;
                        if(w < nt * nt) rest_next = first_two(w, n0, n1);
                        for(int tile = w; tile < nt * nt; tile += NW)
                        {
                            int const tj = tile / nt, ti = tile - tj * nt;
                            int const i0 = 16 * ti, j0 = 16 * tj;
                            int const mr = u - i0 < 16 ? u - i0 : 16, nc = u - j0 < 16 ? u - j0 : 16;
                            auto const raw0 = n0, raw1 = n1;
                            auto rest = rest_next;
                            if(tile + NW < nt * nt) rest_next = first_two(tile + NW, n0, n1);
                            auto acc = tm.tile_zero();
                            tm.tile_mulsub(acc, Lp + p + i0, ldl, Up + j0 * ldu, ldu, mr, nc, p, lane);
                            tm.tile_add(acc, raw0);
                            tm.tile_add(acc, raw1);
                            while(rest)
                            {
                                auto const rq = pull(__builtin_ctzll(rest), i0, j0, mr, nc);
                                rest &= rest - 1;
                                tm.tile_add(acc, rq);
                            }
                            tm.tile_store(acc, Ss + i0 + j0 * u, u, mr, nc, lane);
                        }
                        return;
                    }
                    for(int tile = w; tile < nt * nt; tile += NW)
                    {
                        int const tj = tile / nt, ti = tile - tj * nt;
                        int const i0 = 16 * ti, j0 = 16 * tj;
                        int const mr = u - i0 < 16 ? u - i0 : 16, nc = u - j0 < 16 ? u - j0 : 16;
                        auto acc = tm.tile_zero();
                        if(full) acc = tm.tile_load(Up + p + i0 + j0 * ldu, ldu, mr, nc, lane);
                        else if(chain)
                        {
                            acc = nxt;
                            if(tile + NW < nt * nt) nxt = chain_tile(tile + NW);
                        }
                        else
                        {
                            {
                                for(int ch = ch0; ch < ch1; ++ch)
                                {
                                    int const cc = V.f_child[ch];
                                    int const uc = V.f_u[cc];
                                    double const* Sc = arena + V.f_sptr[cc];
                                    int const* inv = V.f_inv + V.f_inv_off[ch] + p;
                                    tm.tile_foreach(acc, lane,
                                                    [&](int r, int c, double& v)
                                                    {
                                                        if(r < mr && c < nc)
                                                        {
                                                            int const ci = inv[i0 + r], cj = inv[j0 + c];
                                                            if(ci >= 0 && cj >= 0) v += Sc[ci + cj * uc];
                                                        }
                                                    });
                                }
                            }
                        }
                        tm.tile_mulsub(acc, Lp + p + i0, ldl, Up + j0 * ldu, ldu, mr, nc, p, lane);
                        // (keep: the block stays where it is -- the parent, a mode-3 chain link, works on it in place; a tile is read
                        //  and written by the same wavefront, once)
                        if(keep) tm.tile_store(acc, Up + p + i0 + j0 * ldu, ldu, mr, nc, lane);
                        else
                            tm.tile_store(acc, Ss + i0 + j0 * u, u, mr, nc, lane);
                    }
                });
        }
        long long const ck3 = tm.clock();
        PE_MARK("stores");
        if(fuse)
        {
            for(int i = t0; i < p; i += T) w[c0 + i] = g[i];
            double* vs = arena + V.f_sptr[s] + static_cast<long long>(u) * u;
            if(!keep)
                for(int i = t0; i < u; i += T) vs[i] = g[p + i];
        }
        if(CHAIN && cs)
        {
            cs->img = keep ? Up + p : nullptr;
            cs->ld = ldl;
            cs->g = g + p;
        }
        double* Lg = fac + V.f_lptr[s];
        // What later phases read: the backward pass needs U11 (upper triangle of the top p x p block of the L panel) and U12;
        // the sub-diagonal block L21 is read by a separate forward pass only, i.e. when the factors of a linear circuit are
        // reused -- with the fused forward substitution of a non-linear circuit nobody reads it again: not written at all.
        bool const need_l21 = !fuse || V.keep_l21;
        if(!need_l21)
        {
            float const rpp = 1.0f / static_cast<float>(p);
            for(int idx = t0; idx < p * p; idx += T)
            {
                int const k = fdiv(idx, rpp), i = idx - k * p;
                Lg[i + k * m] = Lp[i + k * ldl];
            }
        }
        if(full)
        {
            float const rm = 1.0f / static_cast<float>(m);
            if(need_l21)
                for(int idx = t0; idx < m * p; idx += T)
                {
                    int const k = fdiv(idx, rm), i = idx - k * m;
                    Lg[idx] = Lp[i + k * ldl];
                }
            float const rp = 1.0f / static_cast<float>(p);
            for(int idx = t0; idx < p * u; idx += T)
            {
                int const j = fdiv(idx, rp), r = idx - j * p;
                Lg[m * p + idx] = Up[r + j * ldu];
            }
        }
        else
        {
            // (the panels sit in LDS with odd leading dimensions, in the factor store densely: U panel behind the L panel)
            float const rm = 1.0f / static_cast<float>(m), rp = 1.0f / static_cast<float>(p);
            if(need_l21)
                for(int idx = t0; idx < m * p; idx += T)
                {
                    int const k = fdiv(idx, rm), i = idx - k * m;
                    Lg[idx] = Lp[i + k * ldl];
                }
            for(int idx = t0; idx < p * u; idx += T)
            {
                int const j = fdiv(idx, rp), r = idx - j * p;
                Lg[m * p + idx] = Up[r + j * ldu];
            }
        }
        tm.sync_lds();  // the stores drain behind the next front's loads; readers of S / the panels sit behind a full sync()
        PE_MARK("end");
        if(profile && V.prof && t0 == 0)
        {
            long long* q = V.prof + b * PE_PROF + (profile == 2 ? 32 : 8 + 6 * (full ? 0 : (chain ? 2 : 1)));
            q[0] += ck1 - ck0;
            q[1] += ck2 - ck1;
            q[2] += ck3 - ck2;
            q[3] += tm.clock() - ck3;
            q[4] += 1;
            q[5] += m * m;
            if(profile == 2) q[6] += cka - ck0;  // of the assembly: zeroing + this front's own entries of A
        }
        return true;
    }

    // one PART of the tree (single-workgroup mode: part 0 = everything): wave fronts, then the part's cooperative fronts
    template <class Team>
    PE_DEV bool factor_part(Team const& tm, DevView const& V, int b, int part, double* lds, bool fuse)
    {
        int fail = 0;
        long long const c0 = tm.clock();
        int const* wp = V.wave_ptr + part * (V.n_waves + 1);
        // (quad mode: the wave fronts with V.f_quad were factored by the lane-group kernel, pe_quad.hpp, in the launch before)
        tm.for_each_wave(
            [&](int w, int lane, int NL)
            {
                auto wt = tm.wave_team(lane);
                double* slot = lds + static_cast<long long>(w) * V.lds_slot;
                int const q1 = tm.uniform(wp[w + 1]);
                for(int q = tm.uniform(wp[w]); q < q1; ++q)
                {
                    int const s = tm.uniform(V.wave_list[q]);
                    if(V.quad && V.f_quad[s]) continue;
                    if(!front_factor(wt, V, b, s, slot, V.lds_slot, (w == 0 && part == 0) ? 2 : 0, fuse))
                    {
                        fail = 1;
                        break;
                    }
                }
                // per-wavefront time of the wave phase (load balance of the static subtree assignment), part 0, wavefronts 0..5
                if(V.prof && part == 0 && lane == 0 && w < 6) V.prof[b * PE_PROF + 26 + w] += tm.clock() - c0;
            });
        if(tm.sync_or(fail)) return false;
        long long const c1 = tm.clock();
        for(int q = V.coop_ptr[part]; q < V.coop_ptr[part + 1]; ++q)
        {
            int const s = V.coop_list[q];
            if(V.quad && V.n_mid > 0 && V.f_kind[s] == 3) continue;  // a MID front: factored by the second lane-group launch (pe_quad.hpp)
            if(!front_factor(tm, V, b, s, lds, V.lds_doubles - 2, 1, fuse)) return false;
        }
        if(V.prof && tm.tid() == 0 && part == 0)
        {
            V.prof[b * PE_PROF + 1] += c1 - c0;
            V.prof[b * PE_PROF + 2] += tm.clock() - c1;
        }
        if(V.prof && tm.tid() == 0 && part < 8) V.prof[b * PE_PROF + 40 + part] += tm.clock() - c0;
        return true;
    }

    template <class Team>
    PE_DEV bool factor_all(Team const& tm, DevView const& V, int b, double* lds, bool fuse)
    {
        return factor_part(tm, V, b, 0, lds, fuse);
    }

    // ================================================================================================
    // Triangular solves on the same tree.  Forward: each front adds its children's update vectors (kept in the arena
    // slots the update matrices used during the factorisation), solves with its L11 and passes its own update vector
    // up -- pull-based, race-free, deterministic.  Backward: each front gathers the already final unknowns of its
    // ancestors.  w is the permuted right-hand side / solution.
    // ================================================================================================
    // out[i] (i < nrow) = init[i] - sum_k A[i + k*lda] * x[k]  (k < ncol), A in global memory, x in LDS.
    // All T threads: thread = (row, column chunk); partial sums meet in LDS `part` (>= T doubles).
    template <class Team>
    PE_DEV void team_matvec_sub(Team const& tm, double const* A, int lda, int nrow, int ncol, double const* x, double* part, int t0, int T)
    {
        if(nrow <= 0) return;
        int const chunks = T / nrow > 0 ? (T / nrow < ncol ? T / nrow : (ncol > 0 ? ncol : 1)) : 1;
        float const rr = 1.0f / static_cast<float>(nrow);
        for(int q = t0; q < chunks * nrow; q += T)
        {
            int const c = fdiv(q, rr), i = q - c * nrow;
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            int k = c;
            for(; k + 3 * chunks < ncol; k += 4 * chunks)
            {
                double const a0 = A[i + k * lda], a1 = A[i + (k + chunks) * lda], a2 = A[i + (k + 2 * chunks) * lda], a3 = A[i + (k + 3 * chunks) * lda];
                acc0 += a0 * x[k];
                acc1 += a1 * x[k + chunks];
                acc2 += a2 * x[k + 2 * chunks];
                acc3 += a3 * x[k + 3 * chunks];
            }
            for(; k < ncol; k += chunks) acc0 += A[i + k * lda] * x[k];
            part[q] = (acc0 + acc1) + (acc2 + acc3);
        }
        tm.sync();
        // caller combines: out[i] = init[i] - sum_c part[c * nrow + i]
    }

    // LDS layout of the triangular-solve phase (one team): t[cap_m] | staged block [cap_stage] | partial sums [T].
    // SMALL fronts (m <= 64 and the whole m x p panel fits the staging block) run entirely on one wavefront's
    // registers after ONE round of loads; larger ones stage the p x p pivot block and finish with a team mat-vec.
    template <class Team>
    PE_DEV void front_forward(Team const& tm, DevView const& V, int b, int s_in, double* lds, int cap_m, int cap_stage)
    {
        int const s = tm.uniform(s_in);
        int const c0 = V.f_col0[s], p = V.f_p[s], u = V.f_u[s], m = p + u;
        int const T = tm.size(), t0 = tm.tid();
        double* w = V.w + static_cast<long long>(b) * V.rows;
        double* arena = V.arena + static_cast<long long>(b) * V.arena_doubles;
        double const* Lg = V.factor + static_cast<long long>(b) * V.factor_doubles + V.f_lptr[s];
        double* t = lds;                              // [m]
        double* Lb = lds + cap_m;                     // staged L11 [p x p, ld p] or the whole L panel [m x p, ld m]
        double* part = Lb + cap_stage;                // [T]
        bool const small = m <= 64 && m * p <= cap_stage;
        int const ldb = small ? m : p;
        for(int i = t0; i < m; i += T) t[i] = i < p ? w[c0 + i] : 0.0;
        if(small) stage_copy<4>(Lb, Lg, m * p, t0, T);
        else
        {
            float const rp = 1.0f / static_cast<float>(p);
            for(int idx = t0; idx < p * p; idx += T)
            {
                int const k = fdiv(idx, rp), i = idx - k * p;
                Lb[idx] = Lg[i + k * m];
            }
        }
        tm.sync();
        for(int ch = V.f_child_ptr[s]; ch < V.f_child_ptr[s + 1]; ++ch)
        {
            int const c = V.f_child[ch];
            double const* uc = arena + V.f_sptr[c] + static_cast<long long>(V.f_u[c]) * V.f_u[c];  // the vector sits behind the update matrix
            int const* rel = V.f_rel + V.f_rows_ptr[c];
            for(int i = t0; i < V.f_u[c]; i += T) t[rel[i]] += uc[i];
            tm.sync_lds();
        }
        tm.for_each_wave(
            [&](int w, int lane, int NLw)
            {
                if(w != 0) return;  // wavefront 0 runs the dependent chain in registers, without workgroup barriers
                tm.tri_lower_unit(t, Lb, ldb, p, small ? m : p, lane);
            });
        tm.sync();
        for(int i = t0; i < p; i += T) w[c0 + i] = t[i];
        if(u > 0)
        {
            double* us = arena + V.f_sptr[s] + static_cast<long long>(u) * u;
            if(small)
                for(int i = t0; i < u; i += T) us[i] = t[p + i];
            else
            {
                int const chunks = T / u > 0 ? (T / u < p ? T / u : p) : 1;
                team_matvec_sub(tm, Lg + p, m, u, p, t, part, t0, T);
                for(int i = t0; i < u; i += T)
                {
                    double acc = t[p + i];
                    for(int c = 0; c < chunks; ++c) acc -= part[c * u + i];
                    us[i] = acc;
                }
            }
        }
        tm.sync_lds();  // w / update-vector stores drain behind the next front's loads (its first sync() is a full fence)
    }

    template <class Team>
    PE_DEV void front_backward(Team const& tm, DevView const& V, int b, int s_in, double* lds, int cap_m, int cap_stage)
    {
        int const s = tm.uniform(s_in);
        int const c0 = V.f_col0[s], p = V.f_p[s], u = V.f_u[s], m = p + u;
        int const T = tm.size(), t0 = tm.tid();
        double* w = V.w + static_cast<long long>(b) * V.rows;
        double const* fac = V.factor + static_cast<long long>(b) * V.factor_doubles;
        double const* Lg = fac + V.f_lptr[s];
        double const* Ug = fac + V.f_uptr[s];
        int const* rows = V.f_rows + V.f_rows_ptr[s];
        double* t = lds;                              // [m]
        double* Ub = lds + cap_m;                     // staged U11 [p x p, ld p] (+ U12 [p x u, ld p] behind it when small)
        double* part = Ub + cap_stage;                // [T] partial sums of U12 * x_U
        bool const small = m <= 64 && m * p <= cap_stage;
        for(int j = t0; j < u; j += T) t[p + j] = w[rows[j]];
        for(int i = t0; i < p; i += T) t[i] = w[c0 + i];
        {
            float const rp = 1.0f / static_cast<float>(p);
            for(int idx = t0; idx < p * p; idx += T)
            {
                int const k = fdiv(idx, rp), i = idx - k * p;
                Ub[idx] = Lg[i + k * m];
            }
        }
        if(small) stage_copy<4>(Ub + p * p, Ug, p * u, t0, T);
        tm.sync();
        if(u > 0 && !small)
        {
            int const chunks = T / p > 0 ? (T / p < u ? T / p : u) : 1;
            team_matvec_sub(tm, Ug, p, p, u, t + p, part, t0, T);
            for(int k = t0; k < p; k += T)
            {
                double acc = t[k];
                for(int c = 0; c < chunks; ++c) acc -= part[c * p + k];
                t[k] = acc;
            }
            tm.sync();
        }
        tm.for_each_wave(
            [&](int w, int lane, int NLw)
            {
                if(w != 0) return;
                tm.tri_upper(t, Ub, p, p, small ? u : 0, lane);
            });
        tm.sync();
        for(int i = t0; i < p; i += T) w[c0 + i] = t[i];
        tm.sync();
    }

    // Backward step of a front inside a wavefront's subtree.  A child's update rows are a subset of its parent's rows, so below the
    // subtree's root the ancestors' unknowns come from the parent's solved vector, kept on an LDS stack (offsets fixed by the
    // symbolic analysis) and indexed by the relative map the assembly uses (f_rel) -- not from HBM behind the parent's store.  One memory
    // round per front (its factor panel), no drain between fronts; the root gathers from w as front_backward does.
    template <class Team>
    PE_DEV void front_backward_stacked(Team const& tm, DevView const& V, int b, int s_in, double* sc)
    {
        int const s = tm.uniform(s_in);
        int const c0 = V.f_col0[s], p = V.f_p[s], u = V.f_u[s], m = p + u;
        int const T = tm.size(), t0 = tm.tid();
        double* w = V.w + static_cast<long long>(b) * V.rows;
        double const* fac = V.factor + static_cast<long long>(b) * V.factor_doubles;
        double const* Lg = fac + V.f_lptr[s];
        double const* Ug = fac + V.f_uptr[s];
        int const par = V.f_wpar[s];
        double* stack = sc + V.lds_bstack_off;
        double* t = stack + V.f_wstack[s];  // [m]: this front's solved vector, read by its children
        double* Ub = sc + V.wave_m;        // staged U11 [p x p, ld p] + U12 [p x u, ld p]
        if(par < 0)
        {
            int const* rows = V.f_rows + V.f_rows_ptr[s];
            for(int j = t0; j < u; j += T) t[p + j] = w[rows[j]];
        }
        else
        {
            int const* rel = V.f_rel + V.f_rows_ptr[s];
            double const* tp = stack + par;
            for(int j = t0; j < u; j += T) t[p + j] = tp[rel[j]];
        }
        for(int i = t0; i < p; i += T) t[i] = w[c0 + i];
        {
            float const rp = 1.0f / static_cast<float>(p);
            for(int idx = t0; idx < p * p; idx += T)
            {
                int const k = fdiv(idx, rp), i = idx - k * p;
                Ub[idx] = Lg[i + k * m];
            }
        }
        stage_copy<4>(Ub + p * p, Ug, p * u, t0, T);
        tm.sync_lds();
        tm.for_each_wave(
            [&](int wv, int lane, int)
            {
                if(wv == 0) tm.tri_upper(t, Ub, p, p, u, lane);
            });
        tm.sync_lds();
        for(int i = t0; i < p; i += T) w[c0 + i] = t[i];
    }

    // The same step with a LEAN LDS plan (split schedule: the backward launch then fits 8 workgroups per CU instead of 6, and
    // its 4 096 workgroups run in two rounds instead of three): only U11 (p x p) is staged; the product U12 * x_U reads U12
    // straight from HBM, spread over ALL lanes of the team -- lane -> (row i = lane mod R, column group jg = lane / R), each
    // lane accumulating the columns j = jg, jg + G, ... (G groups of R >= p lanes; 128-byte segments per quarter-wave), the G
    // partial sums per row meet in LDS.  Requires p <= R; columns beyond the first 9 per lane are fetched in further rounds
    // (never on the 64-lane teams the geometry produces: u <= 36).
    template <class Team>
    PE_DEV void front_backward_lean(Team const& tm, DevView const& V, int b, int s_in, double* sc)
    {
        int const s = tm.uniform(s_in);
        int const c0 = V.f_col0[s], p = V.f_p[s], u = V.f_u[s], m = p + u;
        int const T = tm.size(), t0 = tm.tid();
        double* w = V.w + static_cast<long long>(b) * V.rows;
        double const* fac = V.factor + static_cast<long long>(b) * V.factor_doubles;
        double const* Lg = fac + V.f_lptr[s];
        double const* Ug = fac + V.f_uptr[s];
        int const par = V.f_wpar[s];
        double* stack = sc + V.lds_bstack_off_b;
        double* t = stack + V.f_wstack[s];           // [m]: this front's solved vector, read by its children
        double* Ub = sc + V.wave_m;                  // staged U11 [p x p, ld p]
        double* part = Ub + V.lds_wave_stage_b;      // [T] partial sums of U12 * x_U
        // lane -> (row, column group)
        int const R = T >= 64 ? (p <= 16 ? 16 : (p <= 32 ? 32 : 64)) : 1;
        int const G = T / R;
        int const row = t0 % R, jg = t0 / R;
        constexpr int UQ = 9;
        int const nq = (u + G - 1) / G;              // columns per lane
        // global loads first: they depend on nothing in LDS
        double ug[UQ];
        {
            int const i = row < p ? row : p - 1;
#pragma unroll
            for(int q = 0; q < UQ; ++q)
            {
                int const j = jg + G * q;
                ug[q] = (u > 0) ? Ug[i + static_cast<long long>(j < u ? j : u - 1) * p] : 0.0;
            }
        }
        if(par < 0)
        {
            int const* rows = V.f_rows + V.f_rows_ptr[s];
            for(int j = t0; j < u; j += T) t[p + j] = w[rows[j]];
        }
        else
        {
            int const* rel = V.f_rel + V.f_rows_ptr[s];
            double const* tp = stack + par;
            for(int j = t0; j < u; j += T) t[p + j] = tp[rel[j]];
        }
        {
            float const rp = 1.0f / static_cast<float>(p);
            for(int idx = t0; idx < p * p; idx += T)
            {
                int const k = fdiv(idx, rp), i = idx - k * p;
                Ub[idx] = Lg[i + k * m];
            }
        }
        double const wi = t0 < p ? w[c0 + t0] : 0.0;
        tm.sync_lds();
        for(int i = row; i < p; i += R)  // (one pass on a 64-lane team; the serial emulation walks the rows)
        {
            double acc = 0.0;
            if(i == row && row < p)
            {
#pragma unroll
                for(int q = 0; q < UQ; ++q)
                {
                    int const j = jg + G * q;
                    if(q < nq && j < u) acc += ug[q] * t[p + j];
                }
            }
            else
                for(int q = 0; q < (nq < UQ ? nq : UQ); ++q)
                {
                    int const j = jg + G * q;
                    if(j < u) acc += Ug[i + static_cast<long long>(j) * p] * t[p + j];
                }
            for(int q = UQ; q < nq; ++q)
            {
                int const j = jg + G * q;
                if(j < u) acc += Ug[i + static_cast<long long>(j) * p] * t[p + j];
            }
            part[jg * R + i] = acc;
        }
        tm.sync_lds();
        for(int i = t0; i < p; i += T)
        {
            double acc = i == t0 ? wi : w[c0 + i];
            for(int g2 = 0; g2 < G; ++g2) acc -= part[g2 * R + i];
            t[i] = acc;
        }
        tm.sync_lds();
        tm.for_each_wave(
            [&](int wv, int lane, int)
            {
                if(wv == 0) tm.tri_upper(t, Ub, p, p, 0, lane);
            });
        tm.sync_lds();
        for(int i = t0; i < p; i += T) w[c0 + i] = t[i];
    }

    template <class Team>
    PE_DEV void forward_part(Team const& tm, DevView const& V, int b, int part, double* lds)
    {
        int const* wp = V.wave_ptr + part * (V.n_waves + 1);
        // (quad mode: the wave fronts with V.f_quad were factored by the lane-group kernel, pe_quad.hpp, in the launch before)
        tm.for_each_wave(
            [&](int wv, int lane, int NL)
            {
                auto wt = tm.wave_team(lane);
                double* sc = lds + static_cast<long long>(wv) * V.lds_sslot;
                int const q1 = tm.uniform(wp[wv + 1]);
                for(int q = tm.uniform(wp[wv]); q < q1; ++q) front_forward(wt, V, b, V.wave_list[q], sc, V.wave_m, V.lds_wave_stage);
            });
        tm.sync();
        for(int q = V.coop_ptr[part]; q < V.coop_ptr[part + 1]; ++q) front_forward(tm, V, b, V.coop_list[q], lds, V.max_m, V.lds_coop_stage);
    }

    template <class Team>
    PE_DEV void backward_part(Team const& tm, DevView const& V, int b, int part, double* lds)
    {
        int const* wp = V.wave_ptr + part * (V.n_waves + 1);
        long long const cbw0 = tm.clock();
        for(int q = V.coop_ptr[part + 1] - 1; q >= V.coop_ptr[part]; --q) front_backward(tm, V, b, V.coop_list[q], lds, V.max_m, V.lds_coop_stage);
        tm.sync();
        if(V.prof && tm.tid() == 0 && part == 0) V.prof[b * PE_PROF + 3] += tm.clock() - cbw0;  // cooperative part of the backward pass
        tm.for_each_wave(
            [&](int wv, int lane, int NL)
            {
                auto wt = tm.wave_team(lane);
                double* sc = lds + static_cast<long long>(wv) * V.lds_sslot_b;
                int const q0 = tm.uniform(wp[wv]);
                for(int q = tm.uniform(wp[wv + 1]) - 1; q >= q0; --q)
                {
                    int const sq = tm.uniform(V.wave_list[q]);
                    if(V.quad_back && V.f_quad[sq]) continue;  // solved by k_m2_backward_quads in the launch after this one (pe_quad.hpp)
                    int const pq = V.f_p[sq], mq = pq + V.f_u[sq];
                    // (every wave front qualifies by construction; the general routine stays as the fallback, followed by the gather
                    // that puts its solved vector on the stack for the children)
                    if(pq <= 64 && mq <= V.wave_m && pq * pq <= V.lds_wave_stage_b) front_backward_lean(wt, V, b, sq, sc);
                    else
                    {
                        front_backward(wt, V, b, sq, sc, V.wave_m, V.lds_wave_stage_b);
                        double const* wb = V.w + static_cast<long long>(b) * V.rows;
                        double* tq = sc + V.lds_bstack_off_b + V.f_wstack[sq];
                        int const* rows = V.f_rows + V.f_rows_ptr[sq];
                        for(int i = wt.tid(); i < mq; i += wt.size()) tq[i] = i < pq ? wb[V.f_col0[sq] + i] : wb[rows[i - pq]];
                        wt.sync_lds();
                    }
                }
            });
        tm.sync();
    }

    // single-workgroup mode: the whole tree is part 0
    // w = P rhs (row permutation of the static pivoting)
    template <class Team>
    PE_DEV void permute_rhs(Team const& tm, DevView const& V, int b)
    {
        double const* rhs = V.rhs + static_cast<long long>(b) * V.rows;
        double* w = V.w + static_cast<long long>(b) * V.rows;
        for(int k = tm.tid(); k < V.rows; k += tm.size()) w[k] = rhs[V.row_src[k]];
        tm.sync();
    }

    // `forward_done`: the factorisation just ran with the fused forward substitution (w already holds L^-1 P rhs)
    template <class Team>
    PE_DEV void solve_all(Team const& tm, DevView const& V, int b, double* lds, bool forward_done)
    {
        int const T = tm.size(), t0 = tm.tid();
        double* w = V.w + static_cast<long long>(b) * V.rows;
        double* x = V.x + static_cast<long long>(b) * V.rows;
        if(!forward_done) permute_rhs(tm, V, b);
        long long const c0 = tm.clock();
        if(!forward_done) forward_part(tm, V, b, 0, lds);
        backward_part(tm, V, b, 0, lds);
        if(V.prof && t0 == 0)
        {
            V.prof[b * PE_PROF + 5] += tm.clock() - c0;  // backward (+ the separate forward pass of the factor-reuse path)
        }
        for(int k = t0; k < V.rows; k += T) x[V.col_src[k]] = w[k];
        tm.sync();
    }

    // ------------------------------------------------------------------------------------------------
    // Residual safety net of the static-pivot LU (the reference pivots partially: Eigen SparseLU, diagonal threshold 1.0).
    // After a linear solve: r = b - A x row by row on the assembled values (A in original order through slot_e), and the four
    // maxima of the normwise backward error  eta = ||r||_inf / (||A||_inf ||x||_inf + ||b||_inf).  A thread covers rows
    // tid, tid + size, ..; out4 = this thread's partial {max |r_i|, max_i sum_j |a_ij|, max |x_i|, max |b_i|}; rres (may be
    // null) receives r.  Non-finite values count as +inf.
    // ------------------------------------------------------------------------------------------------
    PE_DEV double finite_abs(double v) { return fabs(v) <= 1.7976931348623157e308 ? fabs(v) : __builtin_inf(); }
    template <class Team>
    PE_DEV void residual_norms(Team const& tm, DevView const& V, int b, double* rres, double (&out4)[4])
    {
        double const* a = V.aval + static_cast<long long>(b) * V.nnzA;
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double const* rhs = V.rhs + static_cast<long long>(b) * V.rows;
        double mr = 0.0, ma = 0.0, mx = 0.0, mb = 0.0;
        for(int r = tm.tid(); r < V.rows; r += tm.size())
        {
            double acc = rhs[r], rowsum = 0.0;
            int const e1 = V.csr_rp[r + 1];
            for(int e = V.csr_rp[r]; e < e1; ++e)
            {
                double const av = a[V.slot_e ? V.slot_e[e] : e];
                acc -= av * x[V.csr_ci[e]];
                rowsum += fabs(av);
            }
            if(rres) rres[r] = acc;
            mr = fmax(mr, finite_abs(acc));
            ma = fmax(ma, finite_abs(rowsum));
            mx = fmax(mx, finite_abs(x[r]));
            mb = fmax(mb, finite_abs(rhs[r]));
        }
        out4[0] = mr;
        out4[1] = ma;
        out4[2] = mx;
        out4[3] = mb;
    }
    // Refinement residual of a small-signal AC point (pe_engine_ac.cpp pe_hip_analyze_ac; the complex system in real-equivalent form):
    // r = b0 - A xacc with A as the last stamp assembled it, written into the instance's right-hand-side VALUE SLOTS
    // dv[rhs0 + row] -- the correction solve that follows gathers its right-hand side from there.  Returns this thread's worst
    // componentwise backward error |r_i| / (|b_i| + sum_j |a_ij x_j|).
    template <class Team>
    PE_DEV double ac_residual(Team const& tm, DevView const& V, int b, double const* xacc_all, double const* b0_all, int rhs0)
    {
        double const* a = V.aval + static_cast<long long>(b) * V.nnzA;
        double const* x = xacc_all + static_cast<long long>(b) * V.rows;
        double const* b0 = b0_all + static_cast<long long>(b) * V.rows;
        double* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        double worst = 0.0;
        for(int r = tm.tid(); r < V.rows; r += tm.size())
        {
            double acc = b0[r], mag = fabs(acc);
            int const e1 = V.csr_rp[r + 1];
            for(int e = V.csr_rp[r]; e < e1; ++e)
            {
                double const t = a[V.slot_e ? V.slot_e[e] : e] * x[V.csr_ci[e]];
                acc -= t;
                mag += fabs(t);
            }
            dv[rhs0 + r] = acc;
            worst = fmax(worst, fabs(acc) / (mag > 0.0 ? mag : 1.0));
        }
        return worst;
    }

    PE_DEV double backward_error(double const (&n4)[4])
    {
        double const den = n4[1] * n4[2] + n4[3];
        return den > 0.0 ? n4[0] / den : (n4[0] > 0.0 ? __builtin_inf() : 0.0);
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
    PE_DEV int solve_point(Team const& tm, DevView const& V, int b, int mode, double t, double last_step, bool reuse_factor, double* lds)
    {
        double* x = V.x + static_cast<long long>(b) * V.rows;
        double* xp = V.xprev + static_cast<long long>(b) * V.rows;
        int const iters = V.nonlinear ? V.max_newton : 1;
        for(int it = 0; it < iters; ++it)
        {
            long long const cb0 = tm.clock();
            if(V.nonlinear)
                for(int r = tm.tid(); r < V.rows; r += tm.size()) xp[r] = x[r];
            long long const c0 = tm.clock();
            eval_devices(tm, V, b, mode, t, last_step);
            tm.sync();
            stamp(tm, V, b);
            tm.sync();
            if(V.prof && tm.tid() == 0) V.prof[b * PE_PROF + 0] += tm.clock() - c0;
            if(!reuse_factor)
            {
                permute_rhs(tm, V, b);
                if(!factor_all(tm, V, b, lds, true)) return ST_SINGULAR;  // forward substitution fused into the factorisation
            }
            solve_all(tm, V, b, lds, !reuse_factor);
            long long const cb1 = tm.clock();
            int nonfinite = 0;
            for(int r = tm.tid(); r < V.rows; r += tm.size())
                if(!(fabs(x[r]) <= 1.7976931348623157e308)) nonfinite = 1;
            if(tm.sync_or(nonfinite)) return ST_SINGULAR;
            // Residual safety net: the iterate that is about to be ACCEPTED is checked (a linear solve, or the Newton iterate that passed
            // the convergence test).  The resident kernel only DETECTS an inaccurate solve (and fails the step like a Newton failure);
            // the host then repeats it on the split schedule, whose host-driven loop refines / re-matches (pe_engine_newton.cpp).
            auto accurate = [&]() -> bool
            {
                if(!(V.residual_tol > 0.0)) return true;
                double n4[4];
                residual_norms(tm, V, b, nullptr, n4);
                tm.team_max4(n4, lds);
                return backward_error(n4) <= V.residual_tol;
            };
            if(!V.nonlinear) return accurate() ? 1 : ST_INACCURATE;
            int const viol = tm.sync_or(newton_violations(tm, V, b));
            if(V.prof && tm.tid() == 0) V.prof[b * PE_PROF + 4] += (c0 - cb0) + (tm.clock() - cb1);  // Newton bookkeeping
            if(!viol) return accurate() ? it + 1 : ST_INACCURATE;
        }
        return ST_NO_CONVERGENCE;
    }

    // TR loop of circult::analyze (circuit.h:242-254) for `nsteps` steps of one instance
    template <class Team>
    PE_DEV void tr_steps(Team const& tm, DevView const& V, int b, double dt, int nsteps, bool reuse_factor, double* lds)
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
            int const it = solve_point(tm, V, b, MODE_TR, t, dt, reuse_factor, lds);
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
    PE_DEV void dc_point(Team const& tm, DevView const& V, int b, int mode, double* lds)
    {
        if(V.status[b] != ST_OK) return;
        int const it = solve_point(tm, V, b, mode, V.t_now[b], V.last_step[b], false, lds);
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
