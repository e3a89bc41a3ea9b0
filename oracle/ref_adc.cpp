// oracle/ref_adc.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Config C4 on the REAL reference: the 16-level flash ADC of test/0028.16b_adc/
// adc16_onehot_pe_sim_and_export.cpp:119-216 (Rin 10k, 16 x 1k ladder, Vref 5 V, 15 comparators Ll 0 / Hl 5, 16 OUTPUT
// probes) with the one-hot encoder built from NOT / AND primitives (out[0] = ~cmp[0]; out[i] = cmp[i-1] & ~cmp[i];
// out[15] = cmp[14], :224-239) instead of Verilog; the 7 input samples of :383-391; per sample: set VDC, analyze() (DC),
// digital_clk() x 2.  Prints one JSON document (tests/golden/adc_c4.json via scripts/make_golden.py adc).
#include <cstdio>
#include <vector>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/controller/comparator.h>
#include <phy_engine/model/models/digital/logical/not.h>
#include <phy_engine/model/models/digital/logical/and.h>
#include <phy_engine/model/models/digital/logical/output.h>
#include <phy_engine/netlist/impl.h>

namespace pe = ::phy_engine;
using pe::netlist::add_model;
using pe::netlist::add_to_node;
using pe::netlist::create_node;

int main()
{
    constexpr std::size_t kLevels = 16, kThresholds = 15;
    constexpr double kVref = 5.0, kRin = 10000.0, kRladder = 1000.0;
    pe::circult c{};
    c.set_analyze_type(pe::analyze_type::DC);
    auto& nl = c.get_netlist();
    auto& vin = create_node(nl);
    std::vector<pe::model::node_t*> n_div(kLevels + 1);
    n_div[0] = &nl.ground_node;
    for(std::size_t i = 1; i <= kLevels; ++i) n_div[i] = &create_node(nl);
    {
        auto [rin, p] = add_model(nl, pe::model::resistance{.r = kRin});
        add_to_node(nl, *rin, 0, vin);
        add_to_node(nl, *rin, 1, nl.ground_node);
    }
    {
        auto [vref, p] = add_model(nl, pe::model::VDC{.V = kVref});
        add_to_node(nl, *vref, 0, *n_div[kLevels]);
        add_to_node(nl, *vref, 1, nl.ground_node);
    }
    for(std::size_t i = 1; i <= kLevels; ++i)
    {
        auto [rr, p] = add_model(nl, pe::model::resistance{.r = kRladder});
        add_to_node(nl, *rr, 0, *n_div[i]);
        add_to_node(nl, *rr, 1, *n_div[i - 1]);
    }
    auto [vsrc, vp] = add_model(nl, pe::model::VDC{.V = 0.0});
    add_to_node(nl, *vsrc, 0, vin);
    add_to_node(nl, *vsrc, 1, nl.ground_node);
    std::vector<pe::model::node_t*> cmp_nodes(kThresholds), out_nodes(kLevels), ncmp_nodes(kThresholds);
    for(std::size_t i = 0; i < kThresholds; ++i)
    {
        cmp_nodes[i] = &create_node(nl);
        pe::model::comparator cmp{};
        cmp.Ll = 0.0;
        cmp.Hl = 5.0;
        auto [u, p] = add_model(nl, std::move(cmp));
        add_to_node(nl, *u, 0, vin);
        add_to_node(nl, *u, 1, *n_div[i + 1]);
        add_to_node(nl, *u, 2, *cmp_nodes[i]);
    }
    for(std::size_t i = 0; i < kLevels; ++i)
    {
        out_nodes[i] = &create_node(nl);
        auto [o, p] = add_model(nl, pe::model::OUTPUT{});
        add_to_node(nl, *o, 0, *out_nodes[i]);
    }
    // encoder: ncmp[i] = ~cmp[i]; out[0] = ncmp[0]; out[i] = cmp[i-1] & ncmp[i]; out[15] = cmp[14] (two inverters in series keep it a gate)
    for(std::size_t i = 0; i < kThresholds; ++i)
    {
        ncmp_nodes[i] = (i == 0) ? out_nodes[0] : &create_node(nl);
        auto [g, p] = add_model(nl, pe::model::NOT{});
        add_to_node(nl, *g, 0, *cmp_nodes[i]);
        add_to_node(nl, *g, 1, *ncmp_nodes[i]);
    }
    for(std::size_t i = 1; i < kThresholds; ++i)
    {
        auto [g, p] = add_model(nl, pe::model::AND{});
        add_to_node(nl, *g, 0, *cmp_nodes[i - 1]);
        add_to_node(nl, *g, 1, *ncmp_nodes[i]);
        add_to_node(nl, *g, 2, *out_nodes[i]);
    }
    {
        // out[15] = cmp[14] as AND(cmp[14], cmp[14])
        auto [g, p] = add_model(nl, pe::model::AND{});
        add_to_node(nl, *g, 0, *cmp_nodes[kThresholds - 1]);
        add_to_node(nl, *g, 1, *cmp_nodes[kThresholds - 1]);
        add_to_node(nl, *g, 2, *out_nodes[kLevels - 1]);
    }

    double const samples[7] = {0.0, (1.0 / 16.0) * kVref - 1e-6, (1.0 / 16.0) * kVref + 1e-6, (8.0 / 16.0) * kVref, (15.0 / 16.0) * kVref - 1e-6,
                               (15.0 / 16.0) * kVref + 1e-6, kVref};
    std::printf("{\"samples\": [\n");
    for(int s = 0; s < 7; ++s)
    {
        (void)vsrc->ptr->set_attribute(0, {.d{samples[s]}, .type{pe::model::variant_type::d}});
        bool const ok = c.analyze();
        c.digital_clk();
        c.digital_clk();
        std::printf(" {\"vin\": %.17g, \"ok\": %d, \"v_vin\": %.17g, \"ladder\": [", samples[s], ok ? 1 : 0, vin.node_information.an.voltage.real());
        for(std::size_t i = 1; i <= kLevels; ++i) std::printf("%s%.17g", i > 1 ? ", " : "", n_div[i]->node_information.an.voltage.real());
        std::printf("], \"cmp\": [");
        for(std::size_t i = 0; i < kThresholds; ++i) std::printf("%s%d", i ? ", " : "", static_cast<int>(cmp_nodes[i]->node_information.dn.state));
        std::printf("], \"out\": [");
        for(std::size_t i = 0; i < kLevels; ++i) std::printf("%s%d", i ? ", " : "", static_cast<int>(out_nodes[i]->node_information.dn.state));
        std::printf("]");
        // SURVEY.md 8d C4 "additionally TR dt 1e-6 x 10 steps to honour 'transient'": the same sample through the transient analysis
        // (the ladder is resistive: ten trapezoidal steps must reproduce the DC point), then the digital ticks again
        c.set_analyze_type(pe::analyze_type::TR);
        c.get_analyze_setting().tr.t_step = 1e-6;
        c.get_analyze_setting().tr.t_stop = 1e-5;
        bool const ok_tr = c.analyze();
        c.digital_clk();
        c.digital_clk();
        c.set_analyze_type(pe::analyze_type::DC);
        std::printf(", \"tr\": {\"ok\": %d, \"v_vin\": %.17g, \"ladder\": [", ok_tr ? 1 : 0, vin.node_information.an.voltage.real());
        for(std::size_t i = 1; i <= kLevels; ++i) std::printf("%s%.17g", i > 1 ? ", " : "", n_div[i]->node_information.an.voltage.real());
        std::printf("], \"cmp\": [");
        for(std::size_t i = 0; i < kThresholds; ++i) std::printf("%s%d", i ? ", " : "", static_cast<int>(cmp_nodes[i]->node_information.dn.state));
        std::printf("], \"out\": [");
        for(std::size_t i = 0; i < kLevels; ++i) std::printf("%s%d", i ? ", " : "", static_cast<int>(out_nodes[i]->node_information.dn.state));
        std::printf("]}}%s\n", s < 6 ? "," : "");
    }
    std::printf("]}\n");
    return 0;
}
