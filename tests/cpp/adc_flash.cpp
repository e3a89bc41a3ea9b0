// Config C4: the 16-level flash ADC of test/0028.16b_adc/adc16_onehot_pe_sim_and_export.cpp:119-216 with the
// one-hot encoder built from NOT / AND primitives; 7 input samples (:383-391); per sample set VDC, analyze() (DC),
// digital_clk() x 2.  Prints one JSON document in the format of oracle/ref_adc.cpp (the real reference), which
// tests/test_gpu_cpp_api.py compares bit-exactly (digital) and to 1e-12 (ladder voltages).  Also checks the expected bin
// the way the reference test does (:339-367).
#include <chrono>
#include <cstdio>
#include <vector>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/controller/comparator.h>
#include <phy_engine/model/models/digital/logical/and.h>
#include <phy_engine/model/models/digital/logical/not.h>
#include <phy_engine/model/models/digital/logical/output.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/resistance.h>
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
        auto [g, p] = add_model(nl, pe::model::AND{});
        add_to_node(nl, *g, 0, *cmp_nodes[kThresholds - 1]);
        add_to_node(nl, *g, 1, *cmp_nodes[kThresholds - 1]);
        add_to_node(nl, *g, 2, *out_nodes[kLevels - 1]);
    }

    double const samples[7] = {0.0, (1.0 / 16.0) * kVref - 1e-6, (1.0 / 16.0) * kVref + 1e-6, (8.0 / 16.0) * kVref, (15.0 / 16.0) * kVref - 1e-6,
                               (15.0 / 16.0) * kVref + 1e-6, kVref};
    int rc = 0;
    std::printf("{\"samples\": [\n");
    for(int s = 0; s < 7; ++s)
    {
        pe::model::variant v{};
        v.d = samples[s];
        v.type = pe::model::variant_type::d;
        (void)vsrc->ptr->set_attribute(0, v);
        bool const ok = c.analyze();
        if(!ok)
        {
            std::fprintf(stderr, "adc: analyze failed: %s\n", c.last_error.c_str());
            rc = 10;
        }
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
    // config C4 throughput: the same seven inputs again, 20 rounds, on the now resident circuit -- per sample one
    // set_attribute + analyze() (DC on the device) + two digital ticks on the host; nothing is printed inside the timed loop
    c.mna_mirror_rows = 0;  // (the host copy of the assembled system after every analyze() is a debugging aid: not part of the timed loop)
    auto const t0{std::chrono::steady_clock::now()};
    int timed = 0;
    for(int round = 0; round < 20 && rc == 0; ++round)
        for(int s = 0; s < 7; ++s, ++timed)
        {
            pe::model::variant v{};
            v.d = samples[s];
            v.type = pe::model::variant_type::d;
            (void)vsrc->ptr->set_attribute(0, v);
            if(!c.analyze()) rc = 11;
            c.digital_clk();
            c.digital_clk();
        }
    double const el{std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()};
    std::printf("], \"samples_per_s\": %.6g, \"timing_note\": \"%d samples (7 inputs x 20 rounds) on the resident circuit: set_attribute + analyze() + 2 digital ticks each\"}\n",
                timed / el, timed);
    return rc;
}
