// Mirrors test/0004.solver/dc.cpp: R1 10 ohm, R2 20 ohm, VDC 3 V -> VA = 3, VB = 2, I = 0.1 A (the reference prints them;
// here they are asserted).
#include <cmath>
#include <cstdio>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/netlist/impl.h>

int main()
{
    ::phy_engine::circult c{};
    c.set_analyze_type(::phy_engine::analyze_type::DC);
    auto& nl{c.get_netlist()};
    auto [R1, R1_pos]{add_model(nl, ::phy_engine::model::resistance{.r = 10.0})};
    auto [R2, R2_pos]{add_model(nl, ::phy_engine::model::resistance{.r = 20.0})};
    auto [VDC, VDC_pos]{add_model(nl, ::phy_engine::model::VDC{.V = 3.0})};
    auto& node1{create_node(nl)};
    add_to_node(nl, *R1, 1, node1);
    add_to_node(nl, *R2, 0, node1);
    auto& node2{create_node(nl)};
    add_to_node(nl, *VDC, 0, node2);
    add_to_node(nl, *R1, 0, node2);
    auto& node3{nl.ground_node};
    add_to_node(nl, *VDC, 1, node3);
    add_to_node(nl, *R2, 1, node3);
    if(!c.analyze())
    {
        std::fprintf(stderr, "dc: analyze failed: %s\n", c.last_error.c_str());
        return 1;
    }
    auto const pv{R1->ptr->generate_pin_view()};
    double const va = pv.pins[0].nodes->node_information.an.voltage.real(), vb = pv.pins[1].nodes->node_information.an.voltage.real();
    double const i = -VDC->ptr->generate_branch_view().branches[0].current.real();
    if(std::abs(va - 3.0) > 1e-12 || std::abs(vb - 2.0) > 1e-12 || std::abs(i - 0.1) > 1e-12)
    {
        std::fprintf(stderr, "dc: VA=%.15g VB=%.15g I=%.15g\n", va, vb, i);
        return 1;
    }
    return 0;
}
