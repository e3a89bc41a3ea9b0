// digital_blocks.cpp -- the digital blocks of SURVEY.md 8f rank 4 (loader codes 210-212, 220-233) on the plug-in API, in the
// idiom of the reference's test/0006.digital/digital_blocks_smoke.cpp: INPUT models drive a block, OUTPUT probes read it,
// analyze() once, then one digital_clk() per input vector.  The program prints every probe after every tick as JSON.
// It compiles unchanged against the reference's headers (oracle/Makefile target ref_digital -> tests/golden/digital_blocks.json,
// test infrastructure) and against this repository's host layer (tests/test_digital_blocks.py compares the two bit for bit).
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

#include <phy_engine/phy_engine.h>
#include <phy_engine/model/models/digital/logical/tri_state.h>
#include <phy_engine/model/models/digital/logical/implication.h>
#include <phy_engine/model/models/digital/logical/non_implication.h>
#include <phy_engine/model/models/digital/combinational/half_adder.h>
#include <phy_engine/model/models/digital/combinational/full_adder.h>
#include <phy_engine/model/models/digital/combinational/half_subtractor.h>
#include <phy_engine/model/models/digital/combinational/full_subtractor.h>
#include <phy_engine/model/models/digital/combinational/mul2.h>
#include <phy_engine/model/models/digital/combinational/d_ff.h>
#include <phy_engine/model/models/digital/combinational/t_ff.h>
#include <phy_engine/model/models/digital/combinational/t_bar_ff.h>
#include <phy_engine/model/models/digital/combinational/jk_ff.h>
#include <phy_engine/model/models/digital/combinational/counter4.h>
#include <phy_engine/model/models/digital/combinational/d_latch.h>
#include <phy_engine/model/models/digital/combinational/d_ff_arstn.h>
#include <phy_engine/model/models/digital/combinational/random_generator4.h>
#include <phy_engine/model/models/digital/logical/eight_bit_input.h>
#include <phy_engine/model/models/digital/logical/eight_bit_display.h>
#include <phy_engine/model/models/digital/logical/schmitt_trigger.h>
#include <phy_engine/model/models/digital/logical/resolve2.h>
#include <phy_engine/model/models/digital/logical/case_eq.h>
#include <phy_engine/model/models/digital/logical/is_unknown.h>
#include <phy_engine/model/models/digital/logical/tick_delay.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/resistance.h>

namespace
{
    namespace pe = ::phy_engine;
    using dns = pe::model::digital_node_statement_t;
    using pe::netlist::add_model;
    using pe::netlist::add_to_node;
    using pe::netlist::create_node;

    bool set_input(pe::model::model_base* m, dns s)
    {
        pe::model::variant vi{};
        vi.type = pe::model::variant_type::digital;
        vi.digital = s;
        return m->ptr->set_attribute(0, vi);
    }
    int probe(pe::model::model_base* m)
    {
        auto const v = m->ptr->get_attribute(0);
        return v.type == pe::model::variant_type::digital ? static_cast<int>(v.digital) : -1;
    }

    // deterministic input stream over {L, H, X, Z} with L / H four times as likely (edges matter for the flip-flops)
    struct stream
    {
        std::uint32_t s{12345u};
        dns next()
        {
            s = s * 1664525u + 1013904223u;
            unsigned const r = (s >> 24) % 10u;
            return r < 4 ? dns::false_state : (r < 8 ? dns::true_state : (r == 8 ? dns::indeterminate_state : dns::high_impedence_state));
        }
    };

    bool first = true;

    // one block: n_in INPUTs on pins in_pins[], n_out OUTPUT probes on pins out_pins[]; `ticks` random vectors (the exhaustive
    // {L,H,X,Z}^n_in table first when it is small)
    template <typename M>
    bool run(char const* name, M block, std::vector<int> in_pins, std::vector<int> out_pins, int ticks)
    {
        pe::circult c{};
        c.set_analyze_type(pe::analyze_type::DC);
        auto& nl = c.get_netlist();
        auto [blk, bp] = add_model(nl, std::move(block));
        std::vector<pe::model::model_base*> ins, outs;
        for(int p: in_pins)
        {
            auto [m, mp] = add_model(nl, pe::model::INPUT{.outputA = dns::false_state});
            auto& n = create_node(nl);
            add_to_node(nl, *m, 0, n);
            add_to_node(nl, *blk, static_cast<std::size_t>(p), n);
            ins.push_back(m);
        }
        for(int p: out_pins)
        {
            auto [m, mp] = add_model(nl, pe::model::OUTPUT{});
            auto& n = create_node(nl);
            add_to_node(nl, *blk, static_cast<std::size_t>(p), n);
            add_to_node(nl, *m, 0, n);
            outs.push_back(m);
        }
        if(!c.analyze()) return false;
        std::vector<std::vector<dns>> vecs;
        int const n_in = static_cast<int>(ins.size());
        if(n_in <= 3)
        {
            int total = 1;
            for(int k = 0; k < n_in; ++k) total *= 4;
            for(int code = 0; code < total; ++code)
            {
                std::vector<dns> v(n_in);
                int x = code;
                for(int k = 0; k < n_in; ++k, x /= 4) v[k] = static_cast<dns>(x % 4);
                vecs.push_back(v);
            }
        }
        stream st{};
        for(int t = 0; t < ticks; ++t)
        {
            std::vector<dns> v(n_in);
            for(auto& e: v) e = st.next();
            vecs.push_back(v);
        }
        std::printf("%s\n \"%s\": {\"in\": [", first ? "" : ",", name);
        first = false;
        for(std::size_t t = 0; t < vecs.size(); ++t)
        {
            std::printf("%s[", t ? "," : "");
            for(int k = 0; k < n_in; ++k) std::printf("%s%d", k ? "," : "", static_cast<int>(vecs[t][k]));
            std::printf("]");
        }
        std::printf("], \"out\": [");
        for(std::size_t t = 0; t < vecs.size(); ++t)
        {
            for(int k = 0; k < n_in; ++k)
                if(!set_input(ins[k], vecs[t][k])) return false;
            c.digital_clk();
            std::printf("%s[", t ? "," : "");
            for(std::size_t k = 0; k < outs.size(); ++k) std::printf("%s%d", k ? "," : "", probe(outs[k]));
            std::printf("]");
        }
        std::printf("]}");
        return true;
    }

    // EIGHT_BIT_INPUT (attribute 0 = value) -> 8 probes, and 8 four-state INPUTs -> EIGHT_BIT_DISPLAY (attributes value, unknown_mask)
    bool run_eight_bit()
    {
        pe::circult c{};
        c.set_analyze_type(pe::analyze_type::DC);
        auto& nl = c.get_netlist();
        auto [src, sp] = add_model(nl, pe::model::EIGHT_BIT_INPUT{.value = 0xA5});
        auto [disp, dp] = add_model(nl, pe::model::EIGHT_BIT_DISPLAY{});
        std::vector<pe::model::model_base*> probes, ins;
        for(int k = 0; k < 8; ++k)
        {
            auto [o, op] = add_model(nl, pe::model::OUTPUT{});
            auto& n = create_node(nl);
            add_to_node(nl, *src, static_cast<std::size_t>(k), n);
            add_to_node(nl, *o, 0, n);
            probes.push_back(o);
            auto [i, ip] = add_model(nl, pe::model::INPUT{.outputA = dns::false_state});
            auto& n2 = create_node(nl);
            add_to_node(nl, *i, 0, n2);
            add_to_node(nl, *disp, static_cast<std::size_t>(k), n2);
            ins.push_back(i);
        }
        if(!c.analyze()) return false;
        stream st{};
        std::printf(",\n \"EIGHT_BIT\": {\"in\": [");
        std::vector<std::vector<int>> in_rows, out_rows;
        std::uint32_t lcg = 777u;
        for(int t = 0; t < 120; ++t)
        {
            lcg = lcg * 1664525u + 1013904223u;
            unsigned const val = (t % 7 == 3) ? (in_rows.empty() ? 0u : static_cast<unsigned>(in_rows.back()[0])) : ((lcg >> 16) & 0xFFu);  // sometimes unchanged
            std::vector<int> row{static_cast<int>(val)};
            pe::model::variant vi{};
            vi.type = pe::model::variant_type::ui8;
            vi.ui8 = static_cast<std::uint_least8_t>(val);
            if(!src->ptr->set_attribute(0, vi)) return false;
            for(int k = 0; k < 8; ++k)
            {
                dns const s = st.next();
                row.push_back(static_cast<int>(s));
                if(!set_input(ins[k], s)) return false;
            }
            c.digital_clk();
            std::vector<int> out;
            for(auto* p: probes) out.push_back(probe(p));
            out.push_back(static_cast<int>(disp->ptr->get_attribute(0).ui8));
            out.push_back(static_cast<int>(disp->ptr->get_attribute(1).ui8));
            in_rows.push_back(row);
            out_rows.push_back(out);
        }
        auto dump = [](std::vector<std::vector<int>> const& rows)
        {
            for(std::size_t t = 0; t < rows.size(); ++t)
            {
                std::printf("%s[", t ? "," : "");
                for(std::size_t k = 0; k < rows[t].size(); ++k) std::printf("%s%d", k ? "," : "", rows[t][k]);
                std::printf("]");
            }
        };
        dump(in_rows);
        std::printf("], \"out\": [");
        dump(out_rows);
        std::printf("]}");
        return true;
    }

    // SCHMITT_TRIGGER on an ANALOG input: a VDC through 1 k onto the input node, a DC analysis per sample, one tick per sample;
    // the voltage walks through both thresholds several times (millivolt steps: "in" is the source value in mV)
    bool run_schmitt_analog(char const* name, bool inverted)
    {
        pe::circult c{};
        c.set_analyze_type(pe::analyze_type::DC);
        auto& nl = c.get_netlist();
        auto [v, vp] = add_model(nl, pe::model::VDC{.V = 0.0});
        auto [r, rp] = add_model(nl, pe::model::resistance{.r = 1000.0});
        pe::model::SCHMITT_TRIGGER trig{};
        trig.inverted = inverted;
        auto [sm, smp] = add_model(nl, std::move(trig));
        auto [o, op] = add_model(nl, pe::model::OUTPUT{});
        auto& n_src = create_node(nl);
        auto& n_in = create_node(nl);
        auto& n_out = create_node(nl);
        add_to_node(nl, *v, 0, n_src);
        add_to_node(nl, *v, 1, nl.ground_node);
        add_to_node(nl, *r, 0, n_src);
        add_to_node(nl, *r, 1, n_in);
        add_to_node(nl, *sm, 0, n_in);
        add_to_node(nl, *sm, 1, n_out);
        add_to_node(nl, *o, 0, n_out);
        int const mv[] = {0, 1000, 2000, 3000, 3300, 3400, 3000, 2000, 1700, 1600, 1000, 5000, 1667, 1666, 3333, 3334, 2500, 0, 3334, 1666, 2500, 4000};
        std::printf(",\n \"%s\": {\"in\": [", name);
        for(std::size_t t = 0; t < sizeof(mv) / sizeof(int); ++t) std::printf("%s[%d]", t ? "," : "", mv[t]);
        std::printf("], \"out\": [");
        for(std::size_t t = 0; t < sizeof(mv) / sizeof(int); ++t)
        {
            pe::model::variant vi{};
            vi.type = pe::model::variant_type::d;
            vi.d = mv[t] * 1e-3;
            if(!v->ptr->set_attribute(0, vi)) return false;
            if(!c.analyze()) return false;
            c.digital_clk();
            std::printf("%s[%d,%d]", t ? "," : "", probe(o), static_cast<int>(sm->ptr->get_attribute(3).digital));
        }
        std::printf("]}");
        return true;
    }
}  // namespace

int main()
{
    namespace m = pe::model;
    std::printf("{");
    bool ok = true;
    ok = ok && run("TRI", m::TRI{}, {0, 1}, {2}, 40);
    ok = ok && run("IMP", m::IMP{}, {0, 1}, {2}, 40);
    ok = ok && run("NIMP", m::NIMP{}, {0, 1}, {2}, 40);
    ok = ok && run("HALF_ADDER", m::HALF_ADDER{}, {0, 1}, {2, 3}, 20);
    ok = ok && run("FULL_ADDER", m::FULL_ADDER{}, {0, 1, 2}, {3, 4}, 20);
    ok = ok && run("HALF_SUB", m::HALF_SUB{}, {0, 1}, {2, 3}, 20);
    ok = ok && run("FULL_SUB", m::FULL_SUB{}, {0, 1, 2}, {3, 4}, 20);
    ok = ok && run("MUL2", m::MUL2{}, {0, 1, 2, 3}, {4, 5, 6, 7}, 300);
    ok = ok && run("DFF", m::DFF{}, {0, 1}, {2}, 200);
    ok = ok && run("TFF", m::TFF{}, {0, 1}, {2}, 200);
    ok = ok && run("T_BAR_FF", m::T_BAR_FF{}, {0, 1}, {2}, 200);
    ok = ok && run("JKFF", m::JKFF{}, {0, 1, 2}, {3}, 300);
    ok = ok && run("COUNTER4", m::COUNTER4{}, {4, 5}, {0, 1, 2, 3}, 300);
    ok = ok && run("COUNTER4_free", m::COUNTER4{}, {4}, {0, 1, 2, 3}, 60);  // enable pin left open: counts on every rising edge
    ok = ok && run("DLATCH", m::DLATCH{}, {0, 1}, {2}, 120);
    ok = ok && run("DFF_ARSTN", m::DFF_ARSTN{}, {0, 1, 2}, {3}, 300);
    ok = ok && run("RANDOM_GENERATOR4", m::RANDOM_GENERATOR4{}, {4, 5}, {0, 1, 2, 3}, 300);
    ok = ok && run("RANDOM_GENERATOR4_free", m::RANDOM_GENERATOR4{}, {4}, {0, 1, 2, 3}, 80);  // reset pin left open
    ok = ok && run("SCHMITT_TRIGGER_digital", m::SCHMITT_TRIGGER{}, {0}, {1}, 30);
    ok = ok && run("RESOLVE2", m::RESOLVE2{}, {0, 1}, {2}, 30);
    ok = ok && run("CASE_EQ", m::CASE_EQ{}, {0, 1}, {2}, 30);
    ok = ok && run("IS_UNKNOWN", m::IS_UNKNOWN{}, {0}, {1}, 20);
    ok = ok && run("TICK_DELAY_0", m::TICK_DELAY{0}, {0}, {1}, 40);
    ok = ok && run("TICK_DELAY_1", m::TICK_DELAY{1}, {0}, {1}, 40);
    ok = ok && run("TICK_DELAY_3", m::TICK_DELAY{3}, {0}, {1}, 60);
    ok = ok && run_eight_bit();
    ok = ok && run_schmitt_analog("SCHMITT_TRIGGER_analog", false);
    ok = ok && run_schmitt_analog("SCHMITT_TRIGGER_analog_inverted", true);
    std::printf("\n}\n");
    return ok ? 0 : 1;
}
