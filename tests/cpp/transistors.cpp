// Level-1 MOSFETs and the forward-active BJT through the plug-in API and through the loader's element codes 50-53.
// Known answers are closed-form (square-law saturation current; KCL at the collector), the circuits are this repository's.
#include <cmath>
#include <cstddef>
#include <cstdio>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/VAC.h>
#include <phy_engine/model/models/linear/capacitor.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/model/models/non-linear/BJT_NPN.h>
#include <phy_engine/model/models/non-linear/nmosfet.h>
#include <phy_engine/model/models/non-linear/pmosfet.h>
#include <phy_engine/netlist/impl.h>
#include <phy_engine_dll_api.h>

namespace pm = ::phy_engine::model;

static int failures = 0;
static void expect(char const* what, double got, double want, double tol)
{
    if(!(std::abs(got - want) <= tol))
    {
        std::fprintf(stderr, "transistors: %s = %.15g, expected %.15g (tol %g)\n", what, got, want, tol);
        ++failures;
    }
}

int main()
{
    {   // NMOS in saturation: (Vdd - vd) / Rd = Kp/2 Vov^2 (1 + lambda vd); then Vth raised through set_attribute
        ::phy_engine::circult c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [vdd, p0]{add_model(nl, pm::VDC{.V = 5.0})};
        auto [vg, p1]{add_model(nl, pm::VDC{.V = 2.0})};
        auto [rd, p2]{add_model(nl, pm::resistance{.r = 2000.0})};
        auto [m1, p3]{add_model(nl, pm::nmosfet{.Kp = 2e-3, .lambda = 0.02, .Vth = 1.0})};
        auto& n_dd{create_node(nl)};
        auto& n_g{create_node(nl)};
        auto& n_d{create_node(nl)};
        auto& gnd{nl.ground_node};
        add_to_node(nl, *vdd, 0, n_dd);
        add_to_node(nl, *vdd, 1, gnd);
        add_to_node(nl, *vg, 0, n_g);
        add_to_node(nl, *vg, 1, gnd);
        add_to_node(nl, *rd, 0, n_dd);
        add_to_node(nl, *rd, 1, n_d);
        add_to_node(nl, *m1, 0, n_d);
        add_to_node(nl, *m1, 1, n_g);
        add_to_node(nl, *m1, 2, gnd);
        for(double vth: {1.0, 1.5})
        {
            pm::variant v{};
            v.d = vth;
            v.type = pm::variant_type::d;
            if(!m1->ptr->set_attribute(2, v)) ++failures;
            if(!c.analyze())
            {
                std::fprintf(stderr, "transistors: nmos analyze failed: %s\n", c.last_error.c_str());
                return 1;
            }
            double const vd = n_d.node_information.an.voltage.real(), vov = 2.0 - vth;
            expect("nmos KCL at the drain", (5.0 - vd) / 2000.0, 0.5 * 2e-3 * vov * vov * (1.0 + 0.02 * vd), 1e-9);
            if(!(vd > vov))  // saturation region
            {
                std::fprintf(stderr, "transistors: nmos not saturated: vd=%g vov=%g\n", vd, vov);
                ++failures;
            }
        }
    }
    {   // NPN amplifier in transient: the collector node obeys KCL with Ic = BetaF Ib at every accepted point
        ::phy_engine::circult c{};
        c.set_analyze_type(::phy_engine::analyze_type::TR);
        c.get_analyze_setting().tr.t_step = 1e-6;
        c.get_analyze_setting().tr.t_stop = 2e-4;
        auto& nl{c.get_netlist()};
        auto [vcc, p0]{add_model(nl, pm::VDC{.V = 9.0})};
        auto [rb, p1]{add_model(nl, pm::resistance{.r = 4.7e5})};
        auto [rc, p2]{add_model(nl, pm::resistance{.r = 2.2e3})};
        auto [q1, p3]{add_model(nl, pm::BJT_NPN{.Is = 1e-15, .N = 1.0, .BetaF = 150.0, .Temp = 27.0, .Area = 1.0})};
        auto [vin, p4]{add_model(nl, pm::VAC{.m_Vp = 0.01, .m_omega = 6.283185307179586e4, .m_phase = 0.0})};
        auto [cin, p5]{add_model(nl, pm::capacitor{.m_kZimag = 1e-6})};
        auto& n_cc{create_node(nl)};
        auto& n_b{create_node(nl)};
        auto& n_c{create_node(nl)};
        auto& n_in{create_node(nl)};
        auto& gnd{nl.ground_node};
        add_to_node(nl, *vcc, 0, n_cc);
        add_to_node(nl, *vcc, 1, gnd);
        add_to_node(nl, *rb, 0, n_cc);
        add_to_node(nl, *rb, 1, n_b);
        add_to_node(nl, *rc, 0, n_cc);
        add_to_node(nl, *rc, 1, n_c);
        add_to_node(nl, *q1, 0, n_b);
        add_to_node(nl, *q1, 1, n_c);
        add_to_node(nl, *q1, 2, gnd);
        add_to_node(nl, *vin, 0, n_in);
        add_to_node(nl, *vin, 1, gnd);
        add_to_node(nl, *cin, 0, n_in);
        add_to_node(nl, *cin, 1, n_b);
        if(!c.analyze())
        {
            std::fprintf(stderr, "transistors: npn analyze failed: %s\n", c.last_error.c_str());
            return 1;
        }
        double const vb = n_b.node_information.an.voltage.real(), vc = n_c.node_information.an.voltage.real();
        double const Ut = 1.380650524e-23 * (27.0 + 273.15) / 1.6021765314e-19;
        double const ib = 1e-15 * (std::exp(vb / Ut) - 1.0);
        expect("npn collector KCL", (9.0 - vc) / 2.2e3, 150.0 * ib, 1e-3 * 150.0 * ib + 1e-9);  // Newton stop rule: 1e-3 relative
        // (200 us after a cold start the 1 uF coupling capacitor still holds the base near ground: the stage is barely on)
        if(!(vb > 0.0 && vb < 0.8 && vc > 0.2 && vc <= 9.0))
        {
            std::fprintf(stderr, "transistors: npn operating point vb=%g vc=%g\n", vb, vc);
            ++failures;
        }
    }
    {   // loader codes 52 / 53: an NMOS common-source stage and its PMOS mirror image share the rails; both saturated
        int elements[] = {0, 4, 4, 52, 1, 53, 1};
        double properties[] = {5.0, 2.0, /* NMOS */ 2e-3, 0.02, 1.0, 2000.0, /* PMOS */ 2e-3, 0.02, 1.0, 2000.0};
        int wires[] = {
            1, 0, 4, 0,  // Vdd+ - Rn A
            1, 1, 0, 0,  // Vdd- - gnd
            2, 0, 3, 1,  // Vg+ (2 V) - NMOS G
            2, 1, 0, 0,  // Vg- - gnd
            4, 1, 3, 0,  // Rn B - NMOS D
            3, 2, 0, 0,  // NMOS S - gnd
            5, 2, 1, 0,  // PMOS S - Vdd
            5, 1, 4, 1,  // PMOS G - NMOS drain node (about 2.9 V: Vsg = 2.1 V)
            5, 0, 6, 0,  // PMOS D - Rp A
            6, 1, 0, 0,  // Rp B - gnd
        };
        std::size_t *vec_pos{}, *chunk_pos{}, comp_size{};
        void* c = create_circuit(elements, sizeof(elements) / sizeof(int), wires, sizeof(wires) / sizeof(int), properties, &vec_pos, &chunk_pos, &comp_size);
        if(!c || comp_size != 6)
        {
            std::fprintf(stderr, "transistors: create_circuit: %s\n", phy_engine_last_error());
            return 1;
        }
        if(circuit_set_analyze_type(c, 1 /* DC */) != 0) return 1;
        double voltage[32]{}, current[32]{};
        std::size_t voltage_ord[7]{}, current_ord[7]{}, digital_ord[7]{};
        bool digital[32]{};
        if(analyze_circuit(c, vec_pos, chunk_pos, comp_size, nullptr, nullptr, nullptr, 0, voltage, voltage_ord, current, current_ord, digital, digital_ord) != 0)
        {
            std::fprintf(stderr, "transistors: analyze_circuit: %s\n", phy_engine_last_error());
            return 1;
        }
        // pins per component: Vdd 2, Vg 2, NMOS 3 (D G S), Rn 2, PMOS 3 (D G S), Rp 2
        double const vdn = voltage[4], vdp = voltage[9];
        expect("loader nmos KCL", (5.0 - vdn) / 2000.0, 0.5 * 2e-3 * 1.0 * 1.0 * (1.0 + 0.02 * vdn), 1e-9);
        double const vov_p = (5.0 - vdn) - 1.0, vsd = 5.0 - vdp;
        double const ip = vsd < vov_p ? 2e-3 * (vov_p * vsd - 0.5 * vsd * vsd) * (1.0 + 0.02 * vsd) : 0.5 * 2e-3 * vov_p * vov_p * (1.0 + 0.02 * vsd);
        expect("loader pmos KCL", vdp / 2000.0, ip, 2e-6);   // Newton stop rule 1e-3 relative on voltages
        destroy_circuit(c, vec_pos, chunk_pos);
    }
    if(failures) std::fprintf(stderr, "transistors: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
