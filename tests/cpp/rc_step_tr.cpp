// Mirrors test/0005.models/rc_step_tr.cpp of the reference (same netlist, same assertion |v - (1 - e^-1)| <= 5e-3 after
// 100 trapezoidal steps) against the MI355X host layer: exit 0 = pass.
#include <cmath>
#include <cstdio>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/capacitor.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/netlist/impl.h>

int main()
{
    ::phy_engine::circult c{};
    c.set_analyze_type(::phy_engine::analyze_type::TR);
    auto& setting{c.get_analyze_setting()};
    constexpr double vstep = 1.0, r = 1000.0, cap = 1e-9, tau = r * cap;
    setting.tr.t_step = tau / 100.0;
    setting.tr.t_stop = tau;
    auto& nl{c.get_netlist()};
    auto [v, v_pos]{add_model(nl, ::phy_engine::model::VDC{.V = vstep})};
    auto [rr, rr_pos]{add_model(nl, ::phy_engine::model::resistance{.r = r})};
    auto [cc, cc_pos]{add_model(nl, ::phy_engine::model::capacitor{.m_kZimag = cap})};
    auto& node_src{create_node(nl)};
    auto& node_out{create_node(nl)};
    auto& gnd{nl.ground_node};
    add_to_node(nl, *v, 0, node_src);
    add_to_node(nl, *v, 1, gnd);
    add_to_node(nl, *rr, 0, node_src);
    add_to_node(nl, *rr, 1, node_out);
    add_to_node(nl, *cc, 0, node_out);
    add_to_node(nl, *cc, 1, gnd);
    if(!c.analyze())
    {
        std::fprintf(stderr, "rc_step_tr: analyze failed: %s\n", c.last_error.c_str());
        return 1;
    }
    double const vout{node_out.node_information.an.voltage.real()};
    double const vexp{vstep * (1.0 - std::exp(-1.0))};
    if(std::abs(vout - vexp) > 5e-3)
    {
        std::fprintf(stderr, "rc_step_tr: vout mismatch vout=%.9g vexp=%.9g\n", vout, vexp);
        return 1;
    }
    // the step count is the reference's floating-point loop bound: 100 or 101 steps of tau/100 (circuit.h:242-254)
    if(!(c.tr_duration > 0.99e-6 && c.tr_duration < 1.02e-6)) return 2;
    return 0;
}
