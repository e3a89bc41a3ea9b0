// pe_ac.cpp -- see pe_ac.hpp.  Stamps follow each model's iterate_ac_define (or its iterate_dc_define where the model has
// none: model_refs/base.h:216-232).
#include "pe_ac.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>

namespace pe
{
    namespace
    {
        struct Emit
        {
            int row, col, src;
            bool set;
        };
    }  // namespace

    bool build_ac_circuit(HostCircuit const& hc, AcCircuit& out, OverlaySpec const* overlay)
    {
        out = AcCircuit{};
        auto& ac = out.hc;
        int const N = hc.rows, NN = hc.n_nodes;
        ac.n_nodes = 2 * N;  // every unknown is "a node" for the (unused) convergence test
        ac.n_branches = 0;
        ac.rows = 2 * N;
        ac.batch = hc.batch;
        ac.nonlinear = false;
        ac.map_gen.assign(PE_HIP_KIND_MAX + 1, {});
        auto slot = [&](int kind, int idx)
        {
            out.slots.push_back({kind, idx});
            return DV_FIXED + static_cast<int>(out.slots.size()) - 1;
        };
        std::vector<Emit> ea, eb;
        // complex entry (r, c) += / = (re ? value : j value): the four real cells of the 2 x 2 block
        auto A_re = [&](int r, int c, int dvi, bool neg, bool set)
        {
            if(r < 0 || c < 0) return;
            int const s = (dvi << 1) | (neg ? 1 : 0);
            ea.push_back({r, c, s, set});
            ea.push_back({r + N, c + N, s, set});
        };
        auto A_im = [&](int r, int c, int dvi, bool neg, bool set)
        {
            if(r < 0 || c < 0) return;
            ea.push_back({r, c + N, (dvi << 1) | (neg ? 0 : 1), set});  // -Ai in the upper right block
            ea.push_back({r + N, c, (dvi << 1) | (neg ? 1 : 0), set});  // +Ai in the lower left block
        };
        // a complex '=' on one cell clears BOTH its parts; the pattern keeps explicit zeros for the part that is not written
        auto A_set_re = [&](int r, int c, int dvi, bool neg)
        {
            A_re(r, c, dvi, neg, true);
            if(r >= 0 && c >= 0)
            {
                ea.push_back({r, c + N, -1, true});
                ea.push_back({r + N, c, -1, true});
            }
        };
        auto A_set_im = [&](int r, int c, int dvi, bool neg)
        {
            A_im(r, c, dvi, neg, true);
            if(r >= 0 && c >= 0)
            {
                ea.push_back({r, c, -1, true});
                ea.push_back({r + N, c + N, -1, true});
            }
        };
        auto B_re = [&](int r, int dvi, bool neg, bool set)
        {
            if(r < 0) return;
            eb.push_back({r, 0, (dvi << 1) | (neg ? 1 : 0), set});
            if(set) eb.push_back({r + N, 0, -1, true});
        };
        auto B_im = [&](int r, int dvi, bool neg, bool set)
        {
            if(r < 0) return;
            eb.push_back({r + N, 0, (dvi << 1) | (neg ? 1 : 0), set});
            if(set) eb.push_back({r, 0, -1, true});
        };
        auto G4re = [&](int a, int b, int dvi)
        {
            A_re(a, a, dvi, false, false);
            A_re(a, b, dvi, true, false);
            A_re(b, a, dvi, true, false);
            A_re(b, b, dvi, false, false);
        };
        auto G4im = [&](int a, int b, int dvi)
        {
            A_im(a, a, dvi, false, false);
            A_im(a, b, dvi, true, false);
            A_im(b, a, dvi, true, false);
            A_im(b, b, dvi, false, false);
        };
        auto incidence = [&](int a, int b, int k)
        {
            A_set_re(a, k, DV_ONE, false);
            A_set_re(b, k, DV_ONE, true);
            A_set_re(k, a, DV_ONE, false);
            A_set_re(k, b, DV_ONE, true);
        };
        for(int k = 0; k < hc.n_drives; ++k)  // circuit.h:1015-1022, every analysis type
        {
            A_set_re(hc.drv_node[k], NN + k, DV_ONE, false);
            A_set_re(NN + k, hc.drv_node[k], DV_ONE, false);
            B_re(NN + k, slot(AcSlot::DRIVE, k), false, true);
        }
        if(overlay)  // host-stamped models (circuit.h:1071-1084 calls their iterate_ac hooks): complex += on their registered cells
        {
            for(size_t i = 0; i < overlay->rows.size(); ++i)
            {
                A_re(overlay->rows[i], overlay->cols[i], slot(AcSlot::OV_A_RE, static_cast<int>(i)), false, false);
                A_im(overlay->rows[i], overlay->cols[i], slot(AcSlot::OV_A_IM, static_cast<int>(i)), false, false);
            }
            for(size_t i = 0; i < overlay->rhs_rows.size(); ++i)
            {
                B_re(overlay->rhs_rows[i], slot(AcSlot::OV_B_RE, static_cast<int>(i)), false, false);
                B_im(overlay->rhs_rows[i], slot(AcSlot::OV_B_IM, static_cast<int>(i)), false, false);
            }
        }
        for(int i = 0; i < hc.nR(); ++i) G4re(hc.r_a[i], hc.r_b[i], slot(AcSlot::R_G, i));           // no iterate_ac: DC stamp
        for(int i = 0; i < hc.nC(); ++i) G4im(hc.c_a[i], hc.c_b[i], slot(AcSlot::C_W, i));           // capacitor.h AC: j omega C
        for(int i = 0; i < hc.nL(); ++i)                                                             // inductor.h AC: D = -j omega L
        {
            incidence(hc.l_a[i], hc.l_b[i], hc.l_k[i]);
            A_set_im(hc.l_k[i], hc.l_k[i], slot(AcSlot::L_W, i), true);
        }
        for(int i = 0; i < hc.nVdc(); ++i) incidence(hc.vdc_a[i], hc.vdc_b[i], hc.vdc_k[i]);         // VDC.h AC: E untouched
        for(int i = 0; i < hc.nVac(); ++i)                                                           // VAC.h:117-118,155: E = m_E
        {
            incidence(hc.vac_a[i], hc.vac_b[i], hc.vac_k[i]);
            int const re = slot(AcSlot::VAC_RE, i), im = slot(AcSlot::VAC_IM, i);
            eb.push_back({hc.vac_k[i], 0, re << 1, true});
            eb.push_back({hc.vac_k[i] + N, 0, im << 1, true});
        }
        // IDC.h AC: nothing
        for(int i = 0; i < hc.nD(); ++i)                                                             // PN_junction.h AC: geq + j omega tt geq, no Ieq
        {
            G4re(hc.d_a[i], hc.d_c[i], slot(AcSlot::D_G, i));
            G4im(hc.d_a[i], hc.d_c[i], slot(AcSlot::D_WC, i));
        }
        for(size_t gi = 0; gi < hc.gen.size(); ++gi)
        {
            auto const& d = hc.gen[gi];
            int const *n = d.n, *k = d.k;
            int const g = static_cast<int>(gi);
            switch(d.kind)
            {
                case PE_HIP_IAC:  // IAC.h:116-117,139-140
                {
                    int const re = slot(AcSlot::IAC_RE, g), im = slot(AcSlot::IAC_IM, g);
                    B_re(n[0], re, true, false);
                    B_im(n[0], im, true, false);
                    B_re(n[1], re, false, false);
                    B_im(n[1], im, false, false);
                    break;
                }
                case PE_HIP_VCCS:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_re(n[0], n[2], v, false, false);
                    A_re(n[0], n[3], v, true, false);
                    A_re(n[1], n[2], v, true, false);
                    A_re(n[1], n[3], v, false, false);
                    break;
                }
                case PE_HIP_VCVS:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_set_re(n[0], k[0], DV_ONE, false);
                    A_set_re(n[1], k[0], DV_ONE, true);
                    A_set_re(k[0], n[0], DV_ONE, false);
                    A_set_re(k[0], n[1], DV_ONE, true);
                    A_set_re(k[0], n[2], v, true);
                    A_set_re(k[0], n[3], v, false);
                    break;
                }
                case PE_HIP_CCCS:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_set_re(n[0], k[0], v, false);
                    A_set_re(n[1], k[0], v, true);
                    A_set_re(n[2], k[0], DV_ONE, false);
                    A_set_re(n[3], k[0], DV_ONE, true);
                    A_set_re(k[0], n[2], DV_ONE, false);
                    A_set_re(k[0], n[3], DV_ONE, true);
                    break;
                }
                case PE_HIP_CCVS:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_set_re(n[0], k[0], DV_ONE, false);
                    A_set_re(n[1], k[0], DV_ONE, true);
                    A_set_re(n[2], k[1], DV_ONE, false);
                    A_set_re(n[3], k[1], DV_ONE, true);
                    A_set_re(k[0], n[0], DV_ONE, false);
                    A_set_re(k[0], n[1], DV_ONE, true);
                    A_set_re(k[1], n[2], DV_ONE, false);
                    A_set_re(k[1], n[3], DV_ONE, true);
                    A_set_re(k[0], k[1], v, true);
                    break;
                }
                case PE_HIP_OPAMP:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_set_re(n[2], k[0], DV_ONE, false);
                    A_set_re(n[3], k[0], DV_ONE, true);
                    A_set_re(k[0], n[2], DV_ONE, false);
                    A_set_re(k[0], n[3], DV_ONE, true);
                    A_re(k[0], n[0], v, true, false);
                    A_re(k[0], n[1], v, false, false);
                    break;
                }
                case PE_HIP_XFMR:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_set_re(n[0], k[0], DV_ONE, false);
                    A_set_re(n[1], k[0], DV_ONE, true);
                    A_set_re(k[0], n[0], DV_ONE, false);
                    A_set_re(k[0], n[1], DV_ONE, true);
                    A_set_re(n[2], k[1], DV_ONE, false);
                    A_set_re(n[3], k[1], DV_ONE, true);
                    A_re(k[0], n[2], v, true, false);
                    A_re(k[0], n[3], v, false, false);
                    A_set_re(k[1], k[1], DV_ONE, false);
                    A_set_re(k[1], k[0], v, false);
                    break;
                }
                case PE_HIP_SWITCH:
                    incidence(n[0], n[1], k[0]);
                    A_set_re(k[0], k[0], slot(AcSlot::GEN_STATIC, g), true);
                    break;
                case PE_HIP_RELAY:
                    incidence(n[2], n[3], k[0]);
                    A_set_re(k[0], k[0], slot(AcSlot::RELAY_R, d.aux), true);
                    break;
                case PE_HIP_VGEN:  // generators' iterate_ac: the source is an AC short (E untouched)
                    incidence(n[0], n[1], k[0]);
                    break;
                case PE_HIP_XFMR_CT:
                {
                    int const v = slot(AcSlot::GEN_STATIC, g);
                    A_set_re(n[0], k[0], DV_ONE, false);
                    A_set_re(n[1], k[0], DV_ONE, true);
                    A_set_re(n[2], k[1], DV_ONE, false);
                    A_set_re(n[3], k[1], DV_ONE, true);
                    A_set_re(n[3], k[2], DV_ONE, false);
                    A_set_re(n[4], k[2], DV_ONE, true);
                    A_set_re(k[1], n[2], DV_ONE, false);
                    A_set_re(k[1], n[3], DV_ONE, true);
                    A_re(k[1], n[0], v, true, false);
                    A_re(k[1], n[1], v, false, false);
                    A_set_re(k[2], n[3], DV_ONE, false);
                    A_set_re(k[2], n[4], DV_ONE, true);
                    A_re(k[2], n[0], v, true, false);
                    A_re(k[2], n[1], v, false, false);
                    A_set_re(k[0], k[0], DV_ONE, false);
                    A_set_re(k[0], k[1], v, false);
                    A_set_re(k[0], k[2], v, false);
                    break;
                }
                case PE_HIP_COUPLED_L:  // coupled_inductors.h:124-158: D = -j omega [[L1 M],[M L2]]
                {
                    A_set_re(n[0], k[0], DV_ONE, false);
                    A_set_re(n[1], k[0], DV_ONE, true);
                    A_set_re(n[2], k[1], DV_ONE, false);
                    A_set_re(n[3], k[1], DV_ONE, true);
                    A_set_re(k[0], n[0], DV_ONE, false);
                    A_set_re(k[0], n[1], DV_ONE, true);
                    A_set_re(k[1], n[2], DV_ONE, false);
                    A_set_re(k[1], n[3], DV_ONE, true);
                    int const w11 = slot(AcSlot::KL_W11, g), w12 = slot(AcSlot::KL_W12, g), w22 = slot(AcSlot::KL_W22, g);
                    A_set_im(k[0], k[0], w11, true);
                    A_set_im(k[0], k[1], w12, true);
                    A_set_im(k[1], k[0], w12, true);
                    A_set_im(k[1], k[1], w22, true);
                    break;
                }
                case PE_HIP_NMOS:  // nmosfet.h AC: gds D-S, gm (Vg - Vs), no Ieq
                case PE_HIP_PMOS:
                {
                    int const gds = slot(AcSlot::N3_0, d.aux), gm = slot(AcSlot::N3_1, d.aux);
                    G4re(n[0], n[2], gds);
                    bool const nm = d.kind == PE_HIP_NMOS;
                    A_re(n[0], n[1], gm, !nm, false);
                    A_re(n[0], n[2], gm, nm, false);
                    A_re(n[2], n[1], gm, nm, false);
                    A_re(n[2], n[2], gm, !nm, false);
                    break;
                }
                case PE_HIP_BJT_NPN:  // BJT_NPN.h AC: geq B-E, gm (Vb - Ve) into C-E
                case PE_HIP_BJT_PNP:
                {
                    int const geq = slot(AcSlot::N3_0, d.aux), gm = slot(AcSlot::N3_1, d.aux);
                    G4re(n[0], n[2], geq);
                    if(d.kind == PE_HIP_BJT_NPN)
                    {
                        A_re(n[1], n[0], gm, false, false);
                        A_re(n[1], n[2], gm, true, false);
                        A_re(n[2], n[0], gm, true, false);
                        A_re(n[2], n[2], gm, false, false);
                    }
                    else
                    {
                        A_re(n[2], n[2], gm, false, false);
                        A_re(n[2], n[0], gm, true, false);
                        A_re(n[1], n[2], gm, true, false);
                        A_re(n[1], n[0], gm, false, false);
                    }
                    break;
                }
                default: break;
            }
        }
        for(int nn = 0; nn < NN; ++nn) A_re(nn, nn, DV_GMIN, false, false);  // circuit.h:1107-1110

        ac.dv_len = DV_FIXED + static_cast<int>(out.slots.size());
        // ---- CSR pattern + contribution lists (src -1 = explicit zero written by a complex '=')
        std::int64_t const R = ac.rows;
        std::vector<std::int64_t> keys;
        keys.reserve(ea.size());
        for(auto const& e: ea) keys.push_back(static_cast<std::int64_t>(e.row) * R + e.col);
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        int const nnz = static_cast<int>(keys.size());
        ac.rp.assign(ac.rows + 1, 0);
        ac.ci.resize(nnz);
        for(int s = 0; s < nnz; ++s)
        {
            ++ac.rp[keys[s] / R + 1];
            ac.ci[s] = static_cast<int>(keys[s] % R);
        }
        for(int r = 0; r < ac.rows; ++r) ac.rp[r + 1] += ac.rp[r];
        std::vector<std::vector<int>> per(nnz);
        for(auto const& e: ea)
        {
            std::int64_t const key = static_cast<std::int64_t>(e.row) * R + e.col;
            int const s = static_cast<int>(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
            if(e.set) per[s].clear();
            if(e.src >= 0) per[s].push_back(e.src);
        }
        ac.a_ptr.assign(nnz + 1, 0);
        for(int s = 0; s < nnz; ++s) ac.a_ptr[s + 1] = ac.a_ptr[s] + static_cast<int>(per[s].size());
        ac.a_src.resize(ac.a_ptr[nnz]);
        for(int s = 0; s < nnz; ++s) std::copy(per[s].begin(), per[s].end(), ac.a_src.begin() + ac.a_ptr[s]);
        std::vector<std::vector<int>> perb(ac.rows);
        for(auto const& e: eb)
        {
            if(e.set) perb[e.row].clear();
            if(e.src >= 0) perb[e.row].push_back(e.src);
        }
        ac.b_ptr.assign(ac.rows + 1, 0);
        for(int r = 0; r < ac.rows; ++r) ac.b_ptr[r + 1] = ac.b_ptr[r] + static_cast<int>(perb[r].size());
        ac.b_src.resize(ac.b_ptr[ac.rows]);
        for(int r = 0; r < ac.rows; ++r) std::copy(perb[r].begin(), perb[r].end(), ac.b_src.begin() + ac.b_ptr[r]);
        return true;
    }

    void fill_ac_values(HostCircuit const& hc, AcCircuit const& ac, AcOperatingPoint const& op, int b, double omega, double g_min, double r_open,
                        double* out)
    {
        out[DV_ONE] = 1.0;
        out[DV_GMIN] = g_min;
        size_t const B = static_cast<size_t>(b);
        for(size_t i = 0; i < ac.slots.size(); ++i)
        {
            auto const& s = ac.slots[i];
            double v = 0.0;
            switch(s.kind)
            {
                case AcSlot::R_G: v = hc.r_g[B * hc.nR() + s.idx]; break;
                case AcSlot::C_W: v = omega * hc.c_cap[B * hc.nC() + s.idx]; break;
                case AcSlot::L_W:
                {
                    double const L = hc.l_ind[B * hc.nL() + s.idx];
                    v = (L == 0.0 || omega == 0.0) ? 0.0 : omega * L;
                    break;
                }
                case AcSlot::VAC_RE:
                case AcSlot::VAC_IM:
                {
                    double const* p = &hc.vac_par[(B * hc.nVac() + s.idx) * 3];
                    v = s.kind == AcSlot::VAC_RE ? p[0] * std::cos(p[2]) : p[0] * std::sin(p[2]);
                    break;
                }
                case AcSlot::D_G: v = op.d_geq[B * hc.nD() + s.idx]; break;
                case AcSlot::D_WC:
                {
                    double const geq = op.d_geq[B * hc.nD() + s.idx];
                    double const tt = hc.d_par[(B * hc.nD() + s.idx) * DP_NCOL + DP_TT];
                    double const cd = tt * geq;
                    v = (omega != 0.0 && tt > 0.0 && geq > 0.0 && cd > 0.0) ? cd * omega : 0.0;
                    break;
                }
                case AcSlot::GEN_STATIC:
                {
                    auto const& d = hc.gen[s.idx];
                    (void)gen_static_value(d.kind, &hc.gen_par[B * hc.gen_par_len + d.par], r_open, v);
                    break;
                }
                case AcSlot::RELAY_R: v = op.rl_engaged[B * hc.nRl() + s.idx] ? 0.0 : r_open; break;
                case AcSlot::IAC_RE:
                case AcSlot::IAC_IM:
                {
                    auto const& d = hc.gen[s.idx];
                    double const* p = &hc.gen_par[B * hc.gen_par_len + d.par];
                    v = s.kind == AcSlot::IAC_RE ? p[0] * std::cos(p[2]) : p[0] * std::sin(p[2]);
                    break;
                }
                case AcSlot::KL_W11:
                case AcSlot::KL_W12:
                case AcSlot::KL_W22:
                {
                    auto const& d = hc.gen[s.idx];
                    double const* p = &hc.gen_par[B * hc.gen_par_len + d.par];
                    double const M = p[2] * std::sqrt(p[0] * p[1]);
                    v = omega * (s.kind == AcSlot::KL_W11 ? p[0] : (s.kind == AcSlot::KL_W12 ? M : p[1]));
                    break;
                }
                case AcSlot::N3_0:
                case AcSlot::N3_1:
                {
                    int const kind = hc.n3_kind[s.idx], o = hc.n3_dv[s.idx];
                    bool const mos = kind == PE_HIP_NMOS || kind == PE_HIP_PMOS;
                    // main dv: MOS gds, gm, Ieq | BJT geq, Ieq_be, gm, Ieq_c
                    int const off = s.kind == AcSlot::N3_0 ? 0 : (mos ? 1 : 2);
                    v = op.dv[B * hc.dv_len + o + off];
                    break;
                }
                case AcSlot::DRIVE: v = hc.drv_volt[s.idx]; break;
                case AcSlot::OV_A_RE: v = op.ov_a[s.idx]; break;
                case AcSlot::OV_A_IM: v = op.ov_a[op.ov_a.size() / 2 + s.idx]; break;
                case AcSlot::OV_B_RE: v = op.ov_b[s.idx]; break;
                case AcSlot::OV_B_IM: v = op.ov_b[op.ov_b.size() / 2 + s.idx]; break;
            }
            out[DV_FIXED + i] = v;
        }
    }
}  // namespace pe
