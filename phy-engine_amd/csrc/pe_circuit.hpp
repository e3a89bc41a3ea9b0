// pe_circuit.hpp -- host-side compilation of device tables into the MNA pattern and the contribution lists the
// stamp kernel gathers from.  This is the "pattern discovery" the reference performs implicitly by running every
// model's iterate_* hook into btree_maps (circuits/MNA/mna.h:60-157, circuit.h:993-1003), done once here.
#pragma once
#include <string>
#include <vector>

#include "../../include/pe_hip.h"
#include "pe_device.hpp"

namespace pe
{
    struct HostCircuit
    {
        int n_nodes{}, n_branches{}, n_drives{}, rows{}, batch{1};
        // device arrays; rows are MNA rows (-1 = ground)
        std::vector<int> r_a, r_b;
        std::vector<int> c_a, c_b;
        std::vector<int> l_a, l_b, l_k;
        std::vector<int> vdc_a, vdc_b, vdc_k;
        std::vector<int> vac_a, vac_b, vac_k;
        std::vector<int> idc_a, idc_b;
        std::vector<int> d_a, d_c;
        std::vector<int> drv_node;  // MNA row of each digital drive
        // per-instance parameters [batch][count(*cols)]
        std::vector<double> r_g;      // conductance 1/r
        std::vector<double> c_cap, l_ind, vdc_v, vac_par, idc_i;
        std::vector<double> d_par;    // DP_NCOL derived columns
        std::vector<double> d_raw;    // PE_HIP_DIODE_NPARAM raw columns (kept for pe_hip_update_param)
        std::vector<double> drv_volt; // [n_drives]
        // original table index -> compact index (-1 when the device has an unconnected pin and is skipped)
        std::vector<int> map_r, map_c, map_l, map_vdc, map_vac, map_idc, map_d;
        // dv layout
        int dv_len{};
        int dv_r{}, dv_cg{}, dv_ci{}, dv_lr{}, dv_lu{}, dv_vdc{}, dv_vac{}, dv_idc{}, dv_dg{}, dv_di{}, dv_drv{};
        // MNA pattern (CSR, sorted columns) + contribution lists
        std::vector<int> rp, ci;
        std::vector<int> a_ptr, a_src;
        std::vector<int> b_ptr, b_src;
        bool nonlinear{};
        // host-stamp overlay: dv slots [dv_ova, dv_ova + n_ov_a) add into the cells (ov_rows, ov_cols), [dv_ovb, ..) into the rhs rows
        int n_ov_a{}, n_ov_b{}, dv_ova{}, dv_ovb{};
        std::vector<double> ov_rep;
        std::string error;

        // ---- the remaining linear stampers (PE_HIP_IAC .. PE_HIP_COUPLED_L), kept in one generic list
        struct GenDev
        {
            int kind;
            int n[5];   // MNA rows of the pins (-1 = ground)
            int k[3];   // absolute rows of its branches
            int par;    // offset of its raw parameter columns inside one instance's block of gen_par
            int dv;     // first dv slot (gen_ndv(kind) slots)
            int aux;    // index in ts_* (IAC, VGEN) or cl_* (COUPLED_L), else -1
        };
        std::vector<GenDev> gen;
        int gen_par_len{};             // doubles per instance
        std::vector<double> gen_par;   // [batch][gen_par_len]
        std::vector<std::vector<int>> map_gen;  // [kind] original table index -> index in gen (-1: unconnected pin)
        int dv_gen{};
        std::vector<int> ts_kind, ts_dv;
        std::vector<double> ts_par;    // [batch][nTs][8]
        std::vector<int> cl_n, cl_k, cl_dv;
        std::vector<double> cl_par;    // [batch][nCl][3]
        std::vector<int> rl_n, rl_dv;
        std::vector<double> rl_par;    // [batch][nRl][2]
        int nRl() const { return static_cast<int>(rl_dv.size()); }
        std::vector<int> n3_kind, n3_n, n3_dv;
        std::vector<double> n3_par;    // [batch][nN3][3]
        int nN3() const { return static_cast<int>(n3_kind.size()); }
        int nTs() const { return static_cast<int>(ts_kind.size()); }
        int nCl() const { return static_cast<int>(cl_dv.size()); }

        int nR() const { return static_cast<int>(r_a.size()); }
        int nC() const { return static_cast<int>(c_a.size()); }
        int nL() const { return static_cast<int>(l_a.size()); }
        int nVdc() const { return static_cast<int>(vdc_a.size()); }
        int nVac() const { return static_cast<int>(vac_a.size()); }
        int nIdc() const { return static_cast<int>(idc_a.size()); }
        int nD() const { return static_cast<int>(d_a.size()); }
    };

    // Host-stamp overlay (pe_hip_set_overlay): matrix cells / right-hand-side rows whose values come from the host once per
    // Newton iteration (plug-in models without a device table).  Each entry owns one dv slot and ADDS it to its cell.
    struct OverlaySpec
    {
        std::vector<int> rows, cols;   // absolute MNA indices (0-based: nodes first, then branches)
        std::vector<double> rep;       // representative values for the row matching
        std::vector<int> rhs_rows;
        bool nonlinear{};
        bool empty() const { return rows.empty() && rhs_rows.empty(); }
    };

    // Build from C-ABI tables.  drives: digital_out sources occupying branches [0, n_drives).
    bool build_circuit(int n_nodes, int n_branches, int batch, int n_tables, pe_hip_device_table const* tables, int n_drives, int const* drv_node,
                       double const* drv_volt, HostCircuit& hc, OverlaySpec const* overlay = nullptr);

    // generic kinds: pins, branch rows, raw parameter columns, dv slots
    int gen_pins(int kind);
    int gen_branches(int kind);
    int gen_ncol(int kind);
    int gen_ndv(int kind);
    // value of the (single) static dv slot of a generic device from its raw parameters; false for kinds whose slots are
    // computed on the device (IAC, VGEN, COUPLED_L)
    bool gen_static_value(int kind, double const* raw, double r_open, double& out);
    // (re)derives ts_par / cl_par of generic device g, instance b, from gen_par
    void gen_derive(HostCircuit& hc, int g, int b);

    // PN_junction prepare_foundation (PN_junction.h:296-354) for one diode: raw[PE_HIP_DIODE_NPARAM] -> der[DP_NCOL]
    void diode_derive(double const* raw, double* der);

    // Representative |values| of the A slots of instance 0 for the row matching (TR: capacitors 2C/dt, inductors
    // 2L/dt; DC-like: open / short), diodes at their zero-bias conductance.
    void estimate_values(HostCircuit const& hc, bool tr_mode, double dt, double gmin, double r_open, std::vector<double>& avals);
    // dv entries written from the current iterate x (junction, MOS / BJT, relay values; host-stamp overlay): what changes between the
    // Newton iterations of one solve point
    std::vector<char> dynamic_dv_mask(HostCircuit const& hc);
}  // namespace pe
