// Small-signal AC through the plug-in API (analyze_type::AC / ACOP, single point and logarithmic sweep) and through the
// loader (circuit_set_ac_omega).  Known answers: test/0012.ac/ac_omega.cpp (R-C corner: |v_out| = 1/sqrt 2, phase -45 deg)
// and the closed form 1 / (1 + j omega R C) at every sweep point; a forward-biased diode adds its OP conductance (ACOP).
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdio>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/linear/VAC.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/capacitor.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/model/models/non-linear/PN_junction.h>
#include <phy_engine/netlist/impl.h>
#include <phy_engine_dll_api.h>

namespace pm = ::phy_engine::model;

static int failures = 0;
static void expect(char const* what, double got, double want, double tol)
{
    if(!(std::abs(got - want) <= tol))
    {
        std::fprintf(stderr, "ac_lowpass: %s = %.15g, expected %.15g (tol %g)\n", what, got, want, tol);
        ++failures;
    }
}

int main()
{
    {
        ::phy_engine::circult c{};
        c.set_analyze_type(::phy_engine::analyze_type::AC);
        c.get_analyze_setting().ac.omega = 1000.0;
        auto& nl{c.get_netlist()};
        auto [vac, p0]{add_model(nl, pm::VAC{.m_Vp = 1.0, .m_omega = 1000.0})};
        auto [r1, p1]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto [c1, p2]{add_model(nl, pm::capacitor{.m_kZimag = 1e-6})};
        auto& n_in{create_node(nl)};
        auto& n_out{create_node(nl)};
        auto& gnd{nl.ground_node};
        add_to_node(nl, *vac, 0, n_in);
        add_to_node(nl, *vac, 1, gnd);
        add_to_node(nl, *r1, 0, n_in);
        add_to_node(nl, *r1, 1, n_out);
        add_to_node(nl, *c1, 0, n_out);
        add_to_node(nl, *c1, 1, gnd);
        if(!c.analyze())
        {
            std::fprintf(stderr, "ac_lowpass: analyze failed: %s\n", c.last_error.c_str());
            return 1;
        }
        auto const v{n_out.node_information.an.voltage};
        expect("|v_out| at the corner", std::abs(v), 1.0 / std::sqrt(2.0), 1e-12);
        expect("arg v_out", std::arg(v), -std::atan(1.0), 1e-12);
        // logarithmic sweep over four decades: H = 1 / (1 + j omega R C) at every point
        auto& ac{c.get_analyze_setting().ac};
        ac.sweep = ::phy_engine::analyzer::AC::sweep_type::log;
        ac.omega_start = 10.0;
        ac.omega_stop = 1e5;
        ac.points = 9;
        if(!c.analyze()) return 1;
        auto const& res{c.get_ac_sweep_results()};
        expect("sweep points", static_cast<double>(res.size()), 9.0, 0.0);
        for(auto const& pt: res)
        {
            std::complex<double> const h = 1.0 / std::complex<double>(1.0, pt.omega * 1e-3);
            expect("sweep |H - v_out|", std::abs(pt.x[n_out.node_index] - h), 0.0, 1e-12);
        }
        expect("first omega", res.front().omega, 10.0, 0.0);
        expect("last omega", res.back().omega, 1e5, 1e-6);
    }
    {   // ACOP: 1 V through 1 kOhm into a diode; the AC source sees R and the diode's small-signal conductance I / (N Ut)
        ::phy_engine::circult c{};
        c.set_analyze_type(::phy_engine::analyze_type::ACOP);
        c.get_analyze_setting().ac.omega = 100.0;
        auto& nl{c.get_netlist()};
        auto [vdc, p0]{add_model(nl, pm::VDC{.V = 1.0})};
        auto [vac, p1]{add_model(nl, pm::VAC{.m_Vp = 1e-3, .m_omega = 100.0})};
        auto [r1, p2]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto [d1, p3]{add_model(nl, pm::PN_junction{})};
        auto& n1{create_node(nl)};
        auto& n2{create_node(nl)};
        auto& n3{create_node(nl)};
        auto& gnd{nl.ground_node};
        add_to_node(nl, *vdc, 0, n1);
        add_to_node(nl, *vdc, 1, gnd);
        add_to_node(nl, *vac, 0, n2);
        add_to_node(nl, *vac, 1, n1);
        add_to_node(nl, *r1, 0, n2);
        add_to_node(nl, *r1, 1, n3);
        add_to_node(nl, *d1, 0, n3);
        add_to_node(nl, *d1, 1, gnd);
        if(!c.analyze())
        {
            std::fprintf(stderr, "ac_lowpass: ACOP analyze failed: %s\n", c.last_error.c_str());
            return 1;
        }
        // the node voltages now hold the AC phasors; the OP current follows from the diode law at the OP the engine found:
        // divider ratio v3 / v2 = rd / (R + rd) with rd = N Ut / (Id + Is)  =>  Id = N Ut (1/ratio - 1) / R - Is ~ 0.3-0.45 mA
        double const ratio = (n3.node_information.an.voltage / n2.node_information.an.voltage).real();
        double const Ut = 1.380650524e-23 * (27.0 + 273.15) / 1.6021765314e-19;
        double const id = Ut * (1.0 / ratio - 1.0) / 1000.0;
        if(!(id > 2.5e-4 && id < 5e-4))
        {
            std::fprintf(stderr, "ac_lowpass: ACOP diode current from the small-signal divider = %g A\n", id);
            ++failures;
        }
        expect("ACOP imaginary part (no reactive element)", n3.node_information.an.voltage.imag(), 0.0, 1e-18);
    }
    {   // loader: VAC (code 5: Vp, Hz, deg) - R - C, analyze type 2 (AC) with circuit_set_ac_omega
        int elements[] = {0, 5, 1, 2};
        double properties[] = {1.0, 159.15494309189535, 0.0, 1000.0, 1e-6};
        int wires[] = {1, 0, 2, 0, 1, 1, 0, 0, 2, 1, 3, 0, 3, 1, 0, 0};
        std::size_t *vp{}, *cp{}, cs{};
        void* c = create_circuit(elements, 4, wires, 16, properties, &vp, &cp, &cs);
        if(!c) return 1;
        if(circuit_set_analyze_type(c, 2) != 0 || circuit_set_ac_omega(c, 1000.0) != 0) return 1;
        double voltage[16]{}, current[16]{};
        std::size_t vo[4]{}, co[4]{}, dor[4]{};
        bool digital[16]{};
        if(analyze_circuit(c, vp, cp, cs, nullptr, nullptr, nullptr, 0, voltage, vo, current, co, digital, dor) != 0)
        {
            std::fprintf(stderr, "ac_lowpass: loader: %s\n", phy_engine_last_error());
            return 1;
        }
        expect("loader Re v_out", voltage[vo[2]], 0.5, 1e-12);  // capacitor pin A: 1 / (1 + j) = 0.5 - 0.5 j
        destroy_circuit(c, vp, cp);
    }
    if(failures) std::fprintf(stderr, "ac_lowpass: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
