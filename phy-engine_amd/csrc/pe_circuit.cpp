// pe_circuit.cpp -- see pe_circuit.hpp.
#include "pe_circuit.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>

namespace pe
{
    namespace
    {
        struct Emit
        {
            int row, col, src;
            bool set;  // '=' stamp of the reference (B/C/D/E cells): drops what earlier models wrote to the cell
        };

        inline int row_of(int node_id) { return node_id == 0 ? -1 : node_id - 1; }
    }  // namespace

    int gen_pins(int kind)
    {
        if(kind == PE_HIP_XFMR_CT) return 5;
        if(kind == PE_HIP_RELAY) return 4;
        if(kind >= PE_HIP_NMOS) return 3;
        return (kind == PE_HIP_IAC || kind == PE_HIP_SWITCH || kind == PE_HIP_VGEN) ? 2 : 4;
    }
    int gen_branches(int kind)
    {
        switch(kind)
        {
            case PE_HIP_IAC:
            case PE_HIP_VCCS:
            case PE_HIP_NMOS:
            case PE_HIP_PMOS:
            case PE_HIP_BJT_NPN:
            case PE_HIP_BJT_PNP: return 0;
            case PE_HIP_CCVS:
            case PE_HIP_XFMR:
            case PE_HIP_COUPLED_L: return 2;
            case PE_HIP_XFMR_CT: return 3;
            default: return 1;
        }
    }
    int gen_ncol(int kind)
    {
        switch(kind)
        {
            case PE_HIP_IAC:
            case PE_HIP_COUPLED_L:
            case PE_HIP_NMOS:
            case PE_HIP_PMOS: return 3;
            case PE_HIP_BJT_NPN:
            case PE_HIP_BJT_PNP: return 5;
            case PE_HIP_RELAY: return 2;
            case PE_HIP_VGEN: return PE_HIP_VGEN_NPARAM;
            default: return 1;
        }
    }
    int gen_ndv(int kind)
    {
        if(kind == PE_HIP_COUPLED_L) return 5;
        if(kind == PE_HIP_NMOS || kind == PE_HIP_PMOS) return 3;
        if(kind == PE_HIP_BJT_NPN || kind == PE_HIP_BJT_PNP) return 4;
        return 1;
    }
    bool gen_static_value(int kind, double const* raw, double r_open, double& out)
    {
        switch(kind)
        {
            case PE_HIP_VCCS:
            case PE_HIP_VCVS:
            case PE_HIP_CCCS:
            case PE_HIP_CCVS:
            case PE_HIP_OPAMP:
            case PE_HIP_XFMR: out = raw[0]; return true;
            case PE_HIP_SWITCH: out = raw[0] != 0.0 ? 0.0 : r_open; return true;  // switch.h:93
            case PE_HIP_XFMR_CT: out = raw[0] != 0.0 ? 1.0 / (2.0 * raw[0]) : 0.0; return true;  // 1 / n_half, transformer_center_tap.h:73,108
            default: return false;
        }
    }
    void gen_derive(HostCircuit& hc, int g, int b)
    {
        auto const& d = hc.gen[g];
        double const* raw = &hc.gen_par[static_cast<size_t>(b) * hc.gen_par_len + d.par];
        if(d.kind == PE_HIP_IAC)
        {
            double* o = &hc.ts_par[(static_cast<size_t>(b) * hc.nTs() + d.aux) * 8];
            for(int c = 0; c < 8; ++c) o[c] = c < 3 ? raw[c] : 0.0;
        }
        else if(d.kind == PE_HIP_VGEN)
        {
            double* o = &hc.ts_par[(static_cast<size_t>(b) * hc.nTs() + d.aux) * 8];
            for(int c = 0; c < 8; ++c) o[c] = raw[c];
        }
        else if(d.kind == PE_HIP_COUPLED_L)
        {
            double* o = &hc.cl_par[(static_cast<size_t>(b) * hc.nCl() + d.aux) * 3];
            for(int c = 0; c < 3; ++c) o[c] = raw[c];
        }
        else if(d.kind == PE_HIP_RELAY)
        {
            double* o = &hc.rl_par[(static_cast<size_t>(b) * hc.nRl() + d.aux) * 2];
            o[0] = raw[0];
            o[1] = raw[1];
        }
        else if(d.kind == PE_HIP_NMOS || d.kind == PE_HIP_PMOS)
        {
            double* o = &hc.n3_par[(static_cast<size_t>(b) * hc.nN3() + d.aux) * 3];
            for(int c = 0; c < 3; ++c) o[c] = raw[c];
        }
        else if(d.kind == PE_HIP_BJT_NPN || d.kind == PE_HIP_BJT_PNP)
        {
            // BJT_NPN.h:100-106 (prepare_foundation) + :122-126
            constexpr double kKelvin{-273.15};
            constexpr double qElement{1.6021765314e-19};
            constexpr double kBoltzmann{1.380650524e-23};
            double const Ut = kBoltzmann * (raw[3] - kKelvin) / qElement;
            double* o = &hc.n3_par[(static_cast<size_t>(b) * hc.nN3() + d.aux) * 3];
            o[0] = raw[0] * raw[4];
            o[1] = raw[1] * Ut;
            o[2] = raw[2];
        }
    }

    void diode_derive(double const* raw, double* der)
    {
        // PN_junction.h:296-354
        double const Is = raw[0], N = raw[1], Isr = raw[2], Nr = raw[3], Temp = raw[4], Ibv = raw[5], Bv = raw[6];
        bool const Bv_set = raw[7] != 0.0;
        double const Area = raw[8], tt = raw[9], tt_in_tr = raw[10];
        constexpr double kKelvin{-273.15};
        constexpr double qElement{1.6021765314e-19};
        constexpr double kBoltzmann{1.380650524e-23};
        constexpr double sqrt2{1.4142135623730950488016887242096981};
        double const Is_eff = Is * Area;
        double const Isr_eff = Isr * Area;
        double const Ut = kBoltzmann * (Temp - kKelvin) / qElement;
        double Bv_eff = Bv;
        if(Bv_set) Bv_eff = Bv - N * Ut * std::log(Ibv / Is_eff);
        double const Uth = N * Ut * std::log(N * Ut / (sqrt2 * Is_eff));
        der[DP_IS_EFF] = Is_eff;
        der[DP_ISR_EFF] = Isr_eff;
        der[DP_UTE] = N * Ut;
        der[DP_UTER] = Nr * Ut;
        der[DP_UTH] = Uth;
        der[DP_BV_EFF] = Bv_eff;
        der[DP_BV_SET] = Bv_set ? 1.0 : 0.0;
        der[DP_TT] = tt;
        der[DP_TT_STAMP] = tt_in_tr != 0.0 ? 1.0 : 0.0;
    }

    bool build_circuit(int n_nodes, int n_branches, int batch, int n_tables, pe_hip_device_table const* tables, int n_drives, int const* drv_node,
                       double const* drv_volt, HostCircuit& hc, OverlaySpec const* overlay)
    {
        hc = HostCircuit{};
        if(n_nodes < 0 || n_branches < 0 || batch < 1 || n_drives < 0 || n_drives > n_branches)
        {
            hc.error = "bad circuit dimensions";
            return false;
        }
        hc.n_nodes = n_nodes;
        hc.n_branches = n_branches;
        hc.n_drives = n_drives;
        hc.rows = n_nodes + n_branches;
        hc.batch = batch;
        int const N = n_nodes;

        for(int k = 0; k < n_drives; ++k)
        {
            if(drv_node[k] < 0 || drv_node[k] > n_nodes)
            {
                hc.error = "digital drive node out of range";
                return false;
            }
            hc.drv_node.push_back(row_of(drv_node[k]));
            hc.drv_volt.push_back(drv_volt[k]);
        }

        // ---- pass 1: compact device arrays (devices with an unconnected pin stamp nothing: `if(node_0 && node_1)`)
        hc.map_gen.assign(PE_HIP_KIND_MAX + 1, {});
        std::vector<std::pair<int, int>> gen_src;  // (table, index in table) of every kept generic device
        std::vector<char> branch_used(n_branches, 0);
        for(int k = 0; k < n_drives; ++k) branch_used[k] = 1;
        for(int ti = 0; ti < n_tables; ++ti)
        {
            auto const& t = tables[ti];
            if(t.count < 0 || (t.count > 0 && (!t.nodes || !t.params)))
            {
                hc.error = "device table with null arrays";
                return false;
            }
            if(t.kind >= PE_HIP_IAC && t.kind <= PE_HIP_KIND_MAX)
            {
                int const pins = gen_pins(t.kind), nbr = gen_branches(t.kind), nc = gen_ncol(t.kind);
                if(nbr > 0 && t.count > 0 && !t.branch)
                {
                    hc.error = "branch indices missing for a branch device table";
                    return false;
                }
                auto& map = hc.map_gen[t.kind];
                if(!map.empty())
                {
                    hc.error = "each device kind may appear in one table only";
                    return false;
                }
                for(int i = 0; i < t.count; ++i)
                {
                    bool connected = true;
                    HostCircuit::GenDev d{};
                    d.kind = t.kind;
                    d.aux = -1;
                    for(int q = 0; q < 5; ++q) d.n[q] = -1;
                    for(int q = 0; q < pins; ++q)
                    {
                        int const id = t.nodes[pins * i + q];
                        if(id > n_nodes)
                        {
                            hc.error = "node id out of range";
                            return false;
                        }
                        if(id < 0) connected = false;
                        else
                            d.n[q] = row_of(id);
                    }
                    for(int q = 0; q < nbr; ++q)
                    {
                        int const k = t.branch[nbr * i + q];
                        if(k < n_drives || k >= n_branches || branch_used[k])
                        {
                            hc.error = "branch index out of range or used twice";
                            return false;
                        }
                        branch_used[k] = 1;
                        d.k[q] = N + k;
                    }
                    if(!connected)
                    {
                        map.push_back(-1);
                        continue;
                    }
                    d.par = hc.gen_par_len;
                    hc.gen_par_len += nc;
                    map.push_back(static_cast<int>(hc.gen.size()));
                    hc.gen.push_back(d);
                    gen_src.push_back({ti, i});
                }
                continue;
            }
            int ncol = 1;
            if(t.kind == PE_HIP_VAC) ncol = 3;
            else if(t.kind == PE_HIP_DIODE)
                ncol = PE_HIP_DIODE_NPARAM;
            bool const has_branch = t.kind == PE_HIP_L || t.kind == PE_HIP_VDC || t.kind == PE_HIP_VAC;
            if(has_branch && t.count > 0 && !t.branch)
            {
                hc.error = "branch indices missing for a branch device table";
                return false;
            }
            std::vector<int> keep;
            std::vector<int>* map = nullptr;
            switch(t.kind)
            {
                case PE_HIP_R: map = &hc.map_r; break;
                case PE_HIP_C: map = &hc.map_c; break;
                case PE_HIP_L: map = &hc.map_l; break;
                case PE_HIP_VDC: map = &hc.map_vdc; break;
                case PE_HIP_VAC: map = &hc.map_vac; break;
                case PE_HIP_IDC: map = &hc.map_idc; break;
                case PE_HIP_DIODE: map = &hc.map_d; break;
                default: hc.error = "unknown device kind"; return false;
            }
            if(!map->empty())
            {
                hc.error = "each device kind may appear in one table only";
                return false;
            }
            for(int i = 0; i < t.count; ++i)
            {
                int const a = t.nodes[2 * i], b = t.nodes[2 * i + 1];
                if(a > n_nodes || b > n_nodes)
                {
                    hc.error = "node id out of range";
                    return false;
                }
                if(has_branch)
                {
                    int const k = t.branch[i];
                    if(k < n_drives || k >= n_branches || branch_used[k])
                    {
                        hc.error = "branch index out of range or used twice";
                        return false;
                    }
                    branch_used[k] = 1;
                }
                if(a < 0 || b < 0)
                {
                    map->push_back(-1);
                    continue;
                }
                map->push_back(static_cast<int>(keep.size()));
                keep.push_back(i);
            }
            int const cnt = static_cast<int>(keep.size());
            auto par = [&](int inst, int i, int c) -> double
            { return t.params[(t.params_batched ? static_cast<size_t>(inst) * t.count : 0) * ncol + static_cast<size_t>(i) * ncol + c]; };
            auto fill_nodes = [&](std::vector<int>& A, std::vector<int>& B)
            {
                for(int i: keep)
                {
                    A.push_back(row_of(t.nodes[2 * i]));
                    B.push_back(row_of(t.nodes[2 * i + 1]));
                }
            };
            auto fill_branch = [&](std::vector<int>& K)
            {
                for(int i: keep) K.push_back(N + t.branch[i]);
            };
            auto fill_par1 = [&](std::vector<double>& P, bool invert)
            {
                P.resize(static_cast<size_t>(batch) * cnt);
                for(int inst = 0; inst < batch; ++inst)
                    for(int j = 0; j < cnt; ++j)
                    {
                        double const v = par(inst, keep[j], 0);
                        P[static_cast<size_t>(inst) * cnt + j] = invert ? 1.0 / v : v;
                    }
            };
            switch(t.kind)
            {
                case PE_HIP_R:
                    fill_nodes(hc.r_a, hc.r_b);
                    fill_par1(hc.r_g, true);  // resistance.h:88 `1.0 / r.r`
                    break;
                case PE_HIP_C:
                    fill_nodes(hc.c_a, hc.c_b);
                    fill_par1(hc.c_cap, false);
                    break;
                case PE_HIP_L:
                    fill_nodes(hc.l_a, hc.l_b);
                    fill_branch(hc.l_k);
                    fill_par1(hc.l_ind, false);
                    break;
                case PE_HIP_VDC:
                    fill_nodes(hc.vdc_a, hc.vdc_b);
                    fill_branch(hc.vdc_k);
                    fill_par1(hc.vdc_v, false);
                    break;
                case PE_HIP_VAC:
                    fill_nodes(hc.vac_a, hc.vac_b);
                    fill_branch(hc.vac_k);
                    hc.vac_par.resize(static_cast<size_t>(batch) * cnt * 3);
                    for(int inst = 0; inst < batch; ++inst)
                        for(int j = 0; j < cnt; ++j)
                            for(int c = 0; c < 3; ++c) hc.vac_par[(static_cast<size_t>(inst) * cnt + j) * 3 + c] = par(inst, keep[j], c);
                    break;
                case PE_HIP_IDC:
                    fill_nodes(hc.idc_a, hc.idc_b);
                    fill_par1(hc.idc_i, false);
                    break;
                case PE_HIP_DIODE:
                    fill_nodes(hc.d_a, hc.d_c);
                    hc.d_raw.resize(static_cast<size_t>(batch) * cnt * PE_HIP_DIODE_NPARAM);
                    hc.d_par.resize(static_cast<size_t>(batch) * cnt * DP_NCOL);
                    for(int inst = 0; inst < batch; ++inst)
                        for(int j = 0; j < cnt; ++j)
                        {
                            double* raw = &hc.d_raw[(static_cast<size_t>(inst) * cnt + j) * PE_HIP_DIODE_NPARAM];
                            for(int c = 0; c < PE_HIP_DIODE_NPARAM; ++c) raw[c] = par(inst, keep[j], c);
                            diode_derive(raw, &hc.d_par[(static_cast<size_t>(inst) * cnt + j) * DP_NCOL]);
                        }
                    break;
            }
        }
        hc.nonlinear = hc.nD() > 0;
        // generic devices: raw parameters per instance, auxiliary device arrays
        hc.gen_par.assign(static_cast<size_t>(batch) * hc.gen_par_len, 0.0);
        for(size_t g = 0; g < hc.gen.size(); ++g)
        {
            auto& d = hc.gen[g];
            auto const& t = tables[gen_src[g].first];
            int const i = gen_src[g].second, nc = gen_ncol(d.kind);
            for(int inst = 0; inst < batch; ++inst)
                for(int c = 0; c < nc; ++c)
                    hc.gen_par[static_cast<size_t>(inst) * hc.gen_par_len + d.par + c] =
                        t.params[((t.params_batched ? static_cast<size_t>(inst) * t.count : 0) + static_cast<size_t>(i)) * nc + c];
            if(d.kind == PE_HIP_IAC || d.kind == PE_HIP_VGEN)
            {
                d.aux = hc.nTs();
                int type = 0;
                if(d.kind == PE_HIP_VGEN)
                {
                    double const ty = hc.gen_par[d.par];
                    if(!(ty == 0.0 || ty == 1.0 || ty == 2.0 || ty == 3.0))
                    {
                        hc.error = "generator type must be 0 (sawtooth), 1 (square), 2 (pulse) or 3 (triangle)";
                        return false;
                    }
                    type = 1 + static_cast<int>(ty);
                }
                hc.ts_kind.push_back(type);
                hc.ts_dv.push_back(0);
            }
            else if(d.kind == PE_HIP_RELAY)
            {
                d.aux = hc.nRl();
                hc.rl_n.push_back(d.n[0]);
                hc.rl_n.push_back(d.n[1]);
                hc.rl_dv.push_back(0);
            }
            else if(d.kind == PE_HIP_XFMR_CT) {}
            else if(d.kind >= PE_HIP_NMOS)
            {
                d.aux = hc.nN3();
                hc.n3_kind.push_back(d.kind);
                for(int q = 0; q < 3; ++q) hc.n3_n.push_back(d.n[q]);
                hc.n3_dv.push_back(0);
            }
            else if(d.kind == PE_HIP_COUPLED_L)
            {
                d.aux = hc.nCl();
                for(int q = 0; q < 4; ++q) hc.cl_n.push_back(d.n[q]);
                hc.cl_k.push_back(d.k[0]);
                hc.cl_k.push_back(d.k[1]);
                hc.cl_dv.push_back(0);
            }
        }
        hc.ts_par.assign(static_cast<size_t>(batch) * hc.nTs() * 8, 0.0);
        hc.cl_par.assign(static_cast<size_t>(batch) * hc.nCl() * 3, 0.0);
        hc.n3_par.assign(static_cast<size_t>(batch) * hc.nN3() * 3, 0.0);
        hc.rl_par.assign(static_cast<size_t>(batch) * hc.nRl() * 2, 0.0);
        hc.nonlinear = hc.nonlinear || hc.nN3() > 0 || hc.nRl() > 0;  // relay.h:12: device_type non_linear
        for(size_t g = 0; g < hc.gen.size(); ++g)
            for(int inst = 0; inst < batch; ++inst) gen_derive(hc, static_cast<int>(g), inst);

        // ---- dv layout
        int o = DV_FIXED;
        hc.dv_r = o; o += hc.nR();
        hc.dv_cg = o; o += hc.nC();
        hc.dv_ci = o; o += hc.nC();
        hc.dv_lr = o; o += hc.nL();
        hc.dv_lu = o; o += hc.nL();
        hc.dv_vdc = o; o += hc.nVdc();
        hc.dv_vac = o; o += hc.nVac();
        hc.dv_idc = o; o += hc.nIdc();
        hc.dv_dg = o; o += hc.nD();
        hc.dv_di = o; o += hc.nD();
        hc.dv_drv = o; o += n_drives;
        if(overlay)
        {
            for(size_t i = 0; i < overlay->rows.size(); ++i)
                if(overlay->rows[i] < 0 || overlay->rows[i] >= hc.rows || overlay->cols[i] < 0 || overlay->cols[i] >= hc.rows)
                {
                    hc.error = "overlay cell out of range";
                    return false;
                }
            for(int r: overlay->rhs_rows)
                if(r < 0 || r >= hc.rows)
                {
                    hc.error = "overlay right-hand-side row out of range";
                    return false;
                }
            hc.n_ov_a = static_cast<int>(overlay->rows.size());
            hc.n_ov_b = static_cast<int>(overlay->rhs_rows.size());
            hc.ov_rep = overlay->rep;
            hc.ov_rep.resize(hc.n_ov_a, 1.0);
            hc.nonlinear = hc.nonlinear || overlay->nonlinear;
        }
        hc.dv_ova = o; o += hc.n_ov_a;
        hc.dv_ovb = o; o += hc.n_ov_b;
        hc.dv_gen = o;
        for(auto& d: hc.gen)
        {
            d.dv = o;
            o += gen_ndv(d.kind);
            if(d.kind == PE_HIP_IAC || d.kind == PE_HIP_VGEN) hc.ts_dv[d.aux] = d.dv;
            else if(d.kind == PE_HIP_COUPLED_L)
                hc.cl_dv[d.aux] = d.dv;
            else if(d.kind == PE_HIP_RELAY)
                hc.rl_dv[d.aux] = d.dv;
            else if(d.kind >= PE_HIP_NMOS && d.kind <= PE_HIP_BJT_PNP)
                hc.n3_dv[d.aux] = d.dv;
        }
        hc.dv_len = o;

        // ---- pass 2: emit stamps in the reference's order: digital drives, models, g_min
        std::vector<Emit> ea, eb;
        auto A_add = [&](int r, int c, int dvi, bool neg)
        {
            if(r >= 0 && c >= 0) ea.push_back({r, c, (dvi << 1) | (neg ? 1 : 0), false});
        };
        auto A_set = [&](int r, int c, int dvi, bool neg)
        {
            if(r >= 0 && c >= 0) ea.push_back({r, c, (dvi << 1) | (neg ? 1 : 0), true});
        };
        auto B_add = [&](int r, int dvi, bool neg)
        {
            if(r >= 0) eb.push_back({r, 0, (dvi << 1) | (neg ? 1 : 0), false});
        };
        auto B_set = [&](int r, int dvi, bool neg)
        {
            if(r >= 0) eb.push_back({r, 0, (dvi << 1) | (neg ? 1 : 0), true});
        };
        auto G4 = [&](int a, int b, int dvi)
        {
            A_add(a, a, dvi, false);
            A_add(a, b, dvi, true);
            A_add(b, a, dvi, true);
            A_add(b, b, dvi, false);
        };
        auto incidence = [&](int a, int b, int k)
        {
            A_set(a, k, DV_ONE, false);
            A_set(b, k, DV_ONE, true);
            A_set(k, a, DV_ONE, false);
            A_set(k, b, DV_ONE, true);
        };
        for(int k = 0; k < n_drives; ++k)  // circuit.h:1015-1022
        {
            A_set(hc.drv_node[k], N + k, DV_ONE, false);
            A_set(N + k, hc.drv_node[k], DV_ONE, false);
            B_set(N + k, hc.dv_drv + k, false);
        }
        for(int i = 0; i < hc.nR(); ++i) G4(hc.r_a[i], hc.r_b[i], hc.dv_r + i);
        for(int i = 0; i < hc.nC(); ++i)
        {
            G4(hc.c_a[i], hc.c_b[i], hc.dv_cg + i);
            B_add(hc.c_a[i], hc.dv_ci + i, true);
            B_add(hc.c_b[i], hc.dv_ci + i, false);
        }
        for(int i = 0; i < hc.nL(); ++i)
        {
            incidence(hc.l_a[i], hc.l_b[i], hc.l_k[i]);
            A_set(hc.l_k[i], hc.l_k[i], hc.dv_lr + i, false);
            B_set(hc.l_k[i], hc.dv_lu + i, false);
        }
        for(int i = 0; i < hc.nVdc(); ++i)
        {
            incidence(hc.vdc_a[i], hc.vdc_b[i], hc.vdc_k[i]);
            B_set(hc.vdc_k[i], hc.dv_vdc + i, false);
        }
        for(int i = 0; i < hc.nVac(); ++i)
        {
            incidence(hc.vac_a[i], hc.vac_b[i], hc.vac_k[i]);
            B_set(hc.vac_k[i], hc.dv_vac + i, false);
        }
        for(int i = 0; i < hc.nIdc(); ++i)
        {
            B_add(hc.idc_a[i], hc.dv_idc + i, true);
            B_add(hc.idc_b[i], hc.dv_idc + i, false);
        }
        for(int i = 0; i < hc.nD(); ++i)
        {
            G4(hc.d_a[i], hc.d_c[i], hc.dv_dg + i);
            B_add(hc.d_a[i], hc.dv_di + i, true);
            B_add(hc.d_c[i], hc.dv_di + i, false);
        }
        for(auto const& d: hc.gen)
        {
            int const *n = d.n, *k = d.k, v = d.dv;
            switch(d.kind)
            {
                case PE_HIP_IAC:  // IAC.h:156-157
                    B_add(n[0], v, true);
                    B_add(n[1], v, false);
                    break;
                case PE_HIP_VCCS:  // VCCS.h:89-92
                    A_add(n[0], n[2], v, false);
                    A_add(n[0], n[3], v, true);
                    A_add(n[1], n[2], v, true);
                    A_add(n[1], n[3], v, false);
                    break;
                case PE_HIP_VCVS:  // VCVS.h:92-98
                    A_set(n[0], k[0], DV_ONE, false);
                    A_set(n[1], k[0], DV_ONE, true);
                    A_set(k[0], n[0], DV_ONE, false);
                    A_set(k[0], n[1], DV_ONE, true);
                    A_set(k[0], n[2], v, true);
                    A_set(k[0], n[3], v, false);
                    break;
                case PE_HIP_CCCS:  // CCCS.h:90-95
                    A_set(n[0], k[0], v, false);
                    A_set(n[1], k[0], v, true);
                    A_set(n[2], k[0], DV_ONE, false);
                    A_set(n[3], k[0], DV_ONE, true);
                    A_set(k[0], n[2], DV_ONE, false);
                    A_set(k[0], n[3], DV_ONE, true);
                    break;
                case PE_HIP_CCVS:  // CCVS.h:92-100
                    A_set(n[0], k[0], DV_ONE, false);
                    A_set(n[1], k[0], DV_ONE, true);
                    A_set(n[2], k[1], DV_ONE, false);
                    A_set(n[3], k[1], DV_ONE, true);
                    A_set(k[0], n[0], DV_ONE, false);
                    A_set(k[0], n[1], DV_ONE, true);
                    A_set(k[1], n[2], DV_ONE, false);
                    A_set(k[1], n[3], DV_ONE, true);
                    A_set(k[0], k[1], v, true);
                    break;
                case PE_HIP_OPAMP:  // op_amp.h:74-80
                    A_set(n[2], k[0], DV_ONE, false);
                    A_set(n[3], k[0], DV_ONE, true);
                    A_set(k[0], n[2], DV_ONE, false);
                    A_set(k[0], n[3], DV_ONE, true);
                    A_add(k[0], n[0], v, true);
                    A_add(k[0], n[1], v, false);
                    break;
                case PE_HIP_XFMR:  // transformer.h:80-96 (pins P,Q,S,T; branches kP,kS)
                    A_set(n[0], k[0], DV_ONE, false);
                    A_set(n[1], k[0], DV_ONE, true);
                    A_set(k[0], n[0], DV_ONE, false);
                    A_set(k[0], n[1], DV_ONE, true);
                    A_set(n[2], k[1], DV_ONE, false);
                    A_set(n[3], k[1], DV_ONE, true);
                    A_add(k[0], n[2], v, true);
                    A_add(k[0], n[3], v, false);
                    A_set(k[1], k[1], DV_ONE, false);
                    A_set(k[1], k[0], v, false);
                    break;
                case PE_HIP_SWITCH:  // switch.h:96-100
                    incidence(n[0], n[1], k[0]);
                    A_set(k[0], k[0], v, true);
                    break;
                case PE_HIP_VGEN:  // pulse.h:134-138 & co.
                    incidence(n[0], n[1], k[0]);
                    B_set(k[0], v, false);
                    break;
                case PE_HIP_NMOS:  // nmosfet.h:124-138: gds D-S, gm (Vg - Vs) into D-S, Ieq D->S
                    G4(n[0], n[2], v);
                    A_add(n[0], n[1], v + 1, false);
                    A_add(n[0], n[2], v + 1, true);
                    A_add(n[2], n[1], v + 1, true);
                    A_add(n[2], n[2], v + 1, false);
                    B_add(n[0], v + 2, true);
                    B_add(n[2], v + 2, false);
                    break;
                case PE_HIP_PMOS:  // pmosfet.h:122-137: control is Vs - Vg
                    G4(n[0], n[2], v);
                    A_add(n[0], n[2], v + 1, false);
                    A_add(n[0], n[1], v + 1, true);
                    A_add(n[2], n[2], v + 1, true);
                    A_add(n[2], n[1], v + 1, false);
                    B_add(n[0], v + 2, true);
                    B_add(n[2], v + 2, false);
                    break;
                case PE_HIP_BJT_NPN:  // BJT_NPN.h:135-157 (pins B, C, E)
                    G4(n[0], n[2], v);
                    B_add(n[0], v + 1, true);
                    B_add(n[2], v + 1, false);
                    A_add(n[1], n[0], v + 2, false);
                    A_add(n[1], n[2], v + 2, true);
                    A_add(n[2], n[0], v + 2, true);
                    A_add(n[2], n[2], v + 2, false);
                    B_add(n[1], v + 3, true);
                    B_add(n[2], v + 3, false);
                    break;
                case PE_HIP_BJT_PNP:  // BJT_PNP.h:135-157
                    G4(n[2], n[0], v);
                    B_add(n[2], v + 1, true);
                    B_add(n[0], v + 1, false);
                    A_add(n[2], n[2], v + 2, false);
                    A_add(n[2], n[0], v + 2, true);
                    A_add(n[1], n[2], v + 2, true);
                    A_add(n[1], n[0], v + 2, false);
                    B_add(n[2], v + 3, true);
                    B_add(n[1], v + 3, false);
                    break;
                case PE_HIP_RELAY:  // relay.h:97-102 (pins C+, C-, A, B)
                    incidence(n[2], n[3], k[0]);
                    A_set(k[0], k[0], v, true);
                    break;
                case PE_HIP_XFMR_CT:  // transformer_center_tap.h:96-130 (pins P, Q, S1, CT, S2; branches kP, kH1, kH2); v = 1 / n_half
                    A_set(n[0], k[0], DV_ONE, false);
                    A_set(n[1], k[0], DV_ONE, true);
                    A_set(n[2], k[1], DV_ONE, false);
                    A_set(n[3], k[1], DV_ONE, true);
                    A_set(n[3], k[2], DV_ONE, false);
                    A_set(n[4], k[2], DV_ONE, true);
                    A_set(k[1], n[2], DV_ONE, false);
                    A_set(k[1], n[3], DV_ONE, true);
                    A_add(k[1], n[0], v, true);
                    A_add(k[1], n[1], v, false);
                    A_set(k[2], n[3], DV_ONE, false);
                    A_set(k[2], n[4], DV_ONE, true);
                    A_add(k[2], n[0], v, true);
                    A_add(k[2], n[1], v, false);
                    A_set(k[0], k[0], DV_ONE, false);
                    A_set(k[0], k[1], v, false);
                    A_set(k[0], k[2], v, false);
                    break;
                case PE_HIP_COUPLED_L:  // coupled_inductors.h:223-243 (zeros in the D / E cells reproduce the DC stamp :104-112)
                    A_set(n[0], k[0], DV_ONE, false);
                    A_set(n[1], k[0], DV_ONE, true);
                    A_set(n[2], k[1], DV_ONE, false);
                    A_set(n[3], k[1], DV_ONE, true);
                    A_set(k[0], n[0], DV_ONE, false);
                    A_set(k[0], n[1], DV_ONE, true);
                    A_set(k[1], n[2], DV_ONE, false);
                    A_set(k[1], n[3], DV_ONE, true);
                    A_set(k[0], k[0], v + 0, true);
                    A_set(k[0], k[1], v + 1, true);
                    A_set(k[1], k[0], v + 1, true);
                    A_set(k[1], k[1], v + 2, true);
                    B_set(k[0], v + 3, false);
                    B_set(k[1], v + 4, false);
                    break;
            }
        }
        // host-stamped models (the reference runs them in the same model loop; their cells only ever accumulate)
        for(int i = 0; i < hc.n_ov_a; ++i) A_add(overlay->rows[i], overlay->cols[i], hc.dv_ova + i, false);
        for(int i = 0; i < hc.n_ov_b; ++i) B_add(overlay->rhs_rows[i], hc.dv_ovb + i, false);
        for(int n = 0; n < N; ++n) A_add(n, n, DV_GMIN, false);  // circuit.h:1107-1110

        // ---- CSR pattern
        std::int64_t const R = hc.rows;
        std::vector<std::int64_t> keys;
        keys.reserve(ea.size());
        for(auto const& e: ea) keys.push_back(static_cast<std::int64_t>(e.row) * R + e.col);
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        if(keys.size() > static_cast<size_t>(INT32_MAX))
        {
            hc.error = "matrix exceeds int32 nnz";
            return false;
        }
        int const nnz = static_cast<int>(keys.size());
        hc.rp.assign(hc.rows + 1, 0);
        hc.ci.resize(nnz);
        for(int s = 0; s < nnz; ++s)
        {
            ++hc.rp[keys[s] / R + 1];
            hc.ci[s] = static_cast<int>(keys[s] % R);
        }
        for(int r = 0; r < hc.rows; ++r) hc.rp[r + 1] += hc.rp[r];
        std::vector<std::vector<int>> per(nnz);
        for(auto const& e: ea)
        {
            std::int64_t const key = static_cast<std::int64_t>(e.row) * R + e.col;
            int const s = static_cast<int>(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
            if(e.set) per[s].clear();
            per[s].push_back(e.src);
        }
        hc.a_ptr.assign(nnz + 1, 0);
        for(int s = 0; s < nnz; ++s) hc.a_ptr[s + 1] = hc.a_ptr[s] + static_cast<int>(per[s].size());
        hc.a_src.resize(hc.a_ptr[nnz]);
        for(int s = 0; s < nnz; ++s) std::copy(per[s].begin(), per[s].end(), hc.a_src.begin() + hc.a_ptr[s]);
        std::vector<std::vector<int>> perb(hc.rows);
        for(auto const& e: eb)
        {
            if(e.set) perb[e.row].clear();
            perb[e.row].push_back(e.src);
        }
        hc.b_ptr.assign(hc.rows + 1, 0);
        for(int r = 0; r < hc.rows; ++r) hc.b_ptr[r + 1] = hc.b_ptr[r] + static_cast<int>(perb[r].size());
        hc.b_src.resize(hc.b_ptr[hc.rows]);
        for(int r = 0; r < hc.rows; ++r) std::copy(perb[r].begin(), perb[r].end(), hc.b_src.begin() + hc.b_ptr[r]);
        return true;
    }

    void estimate_values(HostCircuit const& hc, bool tr_mode, double dt, double gmin, double r_open, std::vector<double>& avals)
    {
        std::vector<double> dv(hc.dv_len, 0.0);
        dv[DV_ONE] = 1.0;
        dv[DV_GMIN] = gmin;
        for(int i = 0; i < hc.nR(); ++i) dv[hc.dv_r + i] = hc.r_g[i];
        for(int i = 0; i < hc.n_ov_a; ++i) dv[hc.dv_ova + i] = hc.ov_rep[i];
        bool const dyn = tr_mode && dt > 0.0;
        for(int i = 0; i < hc.nC(); ++i) dv[hc.dv_cg + i] = dyn ? 2.0 * hc.c_cap[i] / dt : 0.0;
        for(int i = 0; i < hc.nL(); ++i) dv[hc.dv_lr + i] = dyn ? -2.0 * hc.l_ind[i] / dt : 0.0;
        for(int i = 0; i < hc.nD(); ++i)
        {
            double const* p = &hc.d_par[static_cast<size_t>(i) * DP_NCOL];
            dv[hc.dv_dg + i] = p[DP_IS_EFF] / p[DP_UTE] + p[DP_ISR_EFF] / p[DP_UTER];
        }
        for(auto const& d: hc.gen)
        {
            double const* raw = &hc.gen_par[d.par];
            double sv;
            if(gen_static_value(d.kind, raw, r_open, sv)) dv[d.dv] = sv;
            else if(d.kind == PE_HIP_RELAY) dv[d.dv] = r_open;
            else if(d.kind == PE_HIP_NMOS || d.kind == PE_HIP_PMOS)
            {
                dv[d.dv] = 1e-2 * raw[0];  // representative gds / gm of a conducting device (|Vov| ~ 10 mV .. 1 V)
                dv[d.dv + 1] = raw[0];
            }
            else if(d.kind == PE_HIP_BJT_NPN || d.kind == PE_HIP_BJT_PNP)
            {
                dv[d.dv] = raw[0] * raw[4] / 0.025;
                dv[d.dv + 2] = raw[2] * dv[d.dv];
            }
            else if(d.kind == PE_HIP_COUPLED_L && dyn)
            {
                double const M = raw[2] * std::sqrt(raw[0] * raw[1]);
                dv[d.dv + 0] = 2.0 * raw[0] / dt;
                dv[d.dv + 1] = 2.0 * M / dt;
                dv[d.dv + 2] = 2.0 * raw[1] / dt;
            }
        }
        int const nnz = static_cast<int>(hc.ci.size());
        avals.assign(nnz, 0.0);
        for(int s = 0; s < nnz; ++s)
        {
            double acc = 0.0;
            for(int e = hc.a_ptr[s]; e < hc.a_ptr[s + 1]; ++e)
            {
                double const v = dv[hc.a_src[e] >> 1];
                acc = (hc.a_src[e] & 1) ? acc - v : acc + v;
            }
            avals[s] = acc;
        }
    }
    std::vector<char> dynamic_dv_mask(HostCircuit const& hc)
    {
        std::vector<char> m(static_cast<size_t>(std::max(hc.dv_len, 1)), 0);
        auto mark = [&](int o, int n)
        {
            for(int k = 0; k < n; ++k)
                if(o + k >= 0 && o + k < hc.dv_len) m[static_cast<size_t>(o + k)] = 1;
        };
        mark(hc.dv_dg, hc.nD());
        mark(hc.dv_di, hc.nD());
        mark(hc.dv_ova, hc.n_ov_a);
        mark(hc.dv_ovb, hc.n_ov_b);
        for(auto const& d: hc.gen)
            if(d.kind == PE_HIP_RELAY || (d.kind >= PE_HIP_NMOS && d.kind <= PE_HIP_BJT_PNP)) mark(d.dv, gen_ndv(d.kind));
        return m;
    }
}  // namespace pe
