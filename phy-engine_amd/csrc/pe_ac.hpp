// pe_ac.hpp -- small-signal AC (circult::run_ac_analysis / solve_once with iterate_ac, circuit.h:389-431,1042-1060) on
// the engine's real kernels: the complex system (Ar + j Ai)(xr + j xi) = br + j bi of N unknowns is solved in its
// real-equivalent form [Ar -Ai; Ai Ar][xr; xi] = [br; bi] of 2N unknowns, so stamping (gather from a per-instance value
// vector), the multifrontal LU and the triangular solves are the ones of the transient path -- only the lists differ.
#pragma once
#include <vector>

#include "pe_circuit.hpp"

namespace pe
{
    // how one slot of the AC value vector is obtained for instance b at angular frequency omega
    struct AcSlot
    {
        enum Kind : int
        {
            R_G,        // conductance of resistor idx
            C_W,        // omega * C of capacitor idx
            L_W,        // omega * L of inductor idx (0 when L == 0 or omega == 0: a short, inductor.h AC stamp)
            VAC_RE,     // Vp cos(phase)
            VAC_IM,     // Vp sin(phase)
            D_G,        // diode idx: small-signal conductance geq of its last linearisation
            D_WC,       // omega * tt * geq (when tt > 0 and geq > 0)
            GEN_STATIC, // generic device idx: its static value (controlled-source gain, 1/n_half, switch contact resistance)
            RELAY_R,    // relay aux idx: contact resistance from its engaged state
            IAC_RE, IAC_IM,      // generic device idx (IAC): Ip cos / sin(phase)
            KL_W11, KL_W12, KL_W22,  // generic device idx (coupled inductors): omega L1, omega M, omega L2
            N3_0, N3_1,  // MOSFET / BJT aux idx: gds, gm | geq, gm of its last linearisation (dv of the main engine)
            DRIVE,      // digital drive idx: its voltage (circuit.h:1015-1022 stamps it in every mode)
            OV_A_RE, OV_A_IM,  // host-stamp overlay cell idx: real / imaginary part of what the models' iterate_ac hooks stamped there
            OV_B_RE, OV_B_IM   // ... right-hand-side row idx
        };
        int kind, idx;
    };

    struct AcCircuit
    {
        HostCircuit hc;             // rows = 2 N, contribution lists only (no device arrays)
        std::vector<AcSlot> slots;  // dv index DV_FIXED + i  <-  slots[i]
    };

    // builds the real-equivalent AC circuit of `hc`
    bool build_ac_circuit(HostCircuit const& hc, AcCircuit& out, OverlaySpec const* overlay = nullptr);

    // main-engine state the AC values depend on, downloaded once per analyze_ac call
    struct AcOperatingPoint
    {
        std::vector<double> d_geq;      // [batch][nD]
        std::vector<double> dv;         // [batch][dv_len] of the main engine (MOSFET / BJT linearisations)
        std::vector<int> rl_engaged;    // [batch][nRl]
        std::vector<double> ov_a, ov_b; // host-stamp overlay at this omega: [n cells] real parts then [n cells] imaginary parts; likewise the rows
    };

    // value vector of instance b at omega: out[dv_len of the AC circuit]
    void fill_ac_values(HostCircuit const& hc, AcCircuit const& ac, AcOperatingPoint const& op, int b, double omega, double g_min, double r_open,
                        double* out);
}  // namespace pe
