// digital_blocks.cpp -- the digital blocks of SURVEY.md 8f rank 4 (loader codes 210-212, 220-229) on the plug-in API, in the
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
    std::printf("\n}\n");
    return ok ? 0 : 1;
}
