// The remaining linear stampers through the plug-in API (same struct / member / free-function names as the reference),
// with the known answers the reference's own model tests assert (test/0005.models/{vccs_dc,vcvs_gain,cccs_dc,ccvs_dc,
// op_amp_follower,transformer_ratio,switch_r_open,generator_dc,coupled_inductors_TR}.cpp).  One table of cases, exit 0 = pass.
#include <cmath>
#include <cstdio>
#include <functional>
#include <numbers>

#include <phy_engine/circuits/circuit.h>
#include <phy_engine/model/models/controller/relay.h>
#include <phy_engine/model/models/controller/switch.h>
#include <phy_engine/model/models/linear/transformer_center_tap.h>
#include <phy_engine/model/models/generator/pulse.h>
#include <phy_engine/model/models/generator/sawtooth.h>
#include <phy_engine/model/models/generator/square.h>
#include <phy_engine/model/models/generator/triangle.h>
#include <phy_engine/model/models/linear/CCCS.h>
#include <phy_engine/model/models/linear/CCVS.h>
#include <phy_engine/model/models/linear/IAC.h>
#include <phy_engine/model/models/linear/VCCS.h>
#include <phy_engine/model/models/linear/VCVS.h>
#include <phy_engine/model/models/linear/VDC.h>
#include <phy_engine/model/models/linear/coupled_inductors.h>
#include <phy_engine/model/models/linear/op_amp.h>
#include <phy_engine/model/models/linear/resistance.h>
#include <phy_engine/model/models/linear/transformer.h>
#include <phy_engine/netlist/impl.h>

namespace pm = ::phy_engine::model;
using circ = ::phy_engine::circult;

namespace
{
    int failures = 0;
    void expect(char const* what, double got, double want, double tol)
    {
        if(!(std::abs(got - want) <= tol))
        {
            std::fprintf(stderr, "linear_models: %s = %.15g, expected %.15g (tol %g)\n", what, got, want, tol);
            ++failures;
        }
    }
    double v(pm::node_t& n) { return n.node_information.an.voltage.real(); }

    // a four-pin controlled element `dev` wired out(S,T) = (out, gnd), control(P,Q) as given
    template <class M>
    void wire4(::phy_engine::netlist::netlist& nl, M* dev, pm::node_t& s, pm::node_t& t, pm::node_t& p, pm::node_t& q)
    {
        add_to_node(nl, *dev, 0, s);
        add_to_node(nl, *dev, 1, t);
        add_to_node(nl, *dev, 2, p);
        add_to_node(nl, *dev, 3, q);
    }
    template <class M>
    void wire2(::phy_engine::netlist::netlist& nl, M* dev, pm::node_t& a, pm::node_t& b)
    {
        add_to_node(nl, *dev, 0, a);
        add_to_node(nl, *dev, 1, b);
    }
    bool run(circ& c, char const* name)
    {
        if(c.analyze()) return true;
        std::fprintf(stderr, "linear_models: %s: analyze failed: %s\n", name, c.last_error.c_str());
        ++failures;
        return false;
    }
}  // namespace

int main()
{
    {   // VCCS: 5 V control, 2 mS into 1 kOhm -> -10 V
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 5.0})};
        auto [gm, p1]{add_model(nl, pm::VCCS{.m_g = 2e-3})};
        auto [rl, p2]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto& ctrl{create_node(nl)};
        auto& out{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, ctrl, gnd);
        wire4(nl, gm, out, gnd, ctrl, gnd);
        wire2(nl, rl, out, gnd);
        if(run(c, "vccs")) expect("vccs vout", v(out), -10.0, 1e-9);
    }
    {   // VCVS gain 2 on 1 V; attribute 0 readable as "Mu"
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 1.0})};
        auto [e, p1]{add_model(nl, pm::VCVS{.m_mu = 2.0})};
        auto [rl, p2]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto& in{create_node(nl)};
        auto& out{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, in, gnd);
        wire4(nl, e, out, gnd, in, gnd);
        wire2(nl, rl, out, gnd);
        if(run(c, "vcvs"))
        {
            expect("vcvs vout", v(out), 2.0, 1e-9);
            expect("vcvs mu attribute", e->ptr->get_attribute(0).d, 2.0, 0.0);
            expect("vcvs branch current", e->ptr->generate_branch_view().branches[0].current.real(), -2e-3, 1e-12);
        }
    }
    {   // CCCS alpha 3 / CCVS r 2k sensing 5 mA
        for(int which = 0; which < 2; ++which)
        {
            circ c{};
            c.set_analyze_type(::phy_engine::analyze_type::DC);
            auto& nl{c.get_netlist()};
            auto [src, p0]{add_model(nl, pm::VDC{.V = 5.0})};
            auto [rin, p1]{add_model(nl, pm::resistance{.r = 1000.0})};
            auto& n_src{create_node(nl)};
            auto& n_ctrl{create_node(nl)};
            auto& n_out{create_node(nl)};
            auto& gnd{nl.ground_node};
            wire2(nl, src, n_src, gnd);
            wire2(nl, rin, n_src, n_ctrl);
            pm::model_base* dev{};
            if(which == 0) dev = add_model(nl, pm::CCCS{.m_alpha = 3.0}).mod;
            else
                dev = add_model(nl, pm::CCVS{.m_r = 2000.0}).mod;
            wire4(nl, dev, n_out, gnd, n_ctrl, gnd);
            auto [rl, p3]{add_model(nl, pm::resistance{.r = 1000.0})};
            wire2(nl, rl, n_out, gnd);
            if(!run(c, which ? "ccvs" : "cccs")) continue;
            auto const bv{dev->ptr->generate_branch_view()};
            if(which == 0)
            {
                expect("cccs branches", static_cast<double>(bv.size), 1.0, 0.0);
                expect("cccs i_ctrl", bv.branches[0].current.real(), 5e-3, 1e-12);
                expect("cccs vout", v(n_out), -15.0, 1e-9);
            }
            else
            {
                expect("ccvs branches", static_cast<double>(bv.size), 2.0, 0.0);
                expect("ccvs i_ctrl", bv.branches[1].current.real(), 5e-3, 1e-12);
                expect("ccvs vout", v(n_out), 10.0, 1e-9);
            }
        }
    }
    {   // op-amp follower, mu = 1e6
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 1.0})};
        auto [oa, p1]{add_model(nl, pm::op_amp{.mu = 1e6})};
        auto [rl, p2]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto& in{create_node(nl)};
        auto& out{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, in, gnd);
        wire4(nl, oa, in, out, out, gnd);
        wire2(nl, rl, out, gnd);
        if(run(c, "op_amp"))
        {
            expect("op_amp vout", v(out), 1e6 / (1.0 + 1e6), 1e-9);
            expect("op_amp |i_branch|", std::abs(oa->ptr->generate_branch_view().branches[0].current.real()), v(out) / 1000.0, 1e-9);
        }
    }
    {   // ideal transformer n = 2: 4 V -> 2 V on 100 Ohm, Is + n Ip = 0
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 4.0})};
        auto [tx, p1]{add_model(nl, pm::transformer{.n = 2.0})};
        auto [rl, p2]{add_model(nl, pm::resistance{.r = 100.0})};
        auto& np{create_node(nl)};
        auto& ns{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, np, gnd);
        wire4(nl, tx, np, gnd, ns, gnd);
        wire2(nl, rl, ns, gnd);
        if(run(c, "transformer"))
        {
            auto const bv{tx->ptr->generate_branch_view()};
            expect("transformer vs", v(ns), 2.0, 1e-9);
            expect("transformer Is + n Ip", bv.branches[1].current.real() + 2.0 * bv.branches[0].current.real(), 0.0, 1e-12);
            expect("transformer |Is|", std::abs(bv.branches[1].current.real()), 0.02, 1e-12);
        }
    }
    {   // open switch leaks 1 / (r_open + R); closing it through set_attribute on the SAME circuit gives the full voltage
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        c.get_environment().r_open = 1e6;
        auto& nl{c.get_netlist()};
        auto [sw, p0]{add_model(nl, pm::single_pole_switch{.cut_through{false}})};
        auto [rl, p1]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto [src, p2]{add_model(nl, pm::VDC{.V = 1.0})};
        auto& a{create_node(nl)};
        auto& b{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, a, gnd);
        wire2(nl, sw, a, b);
        wire2(nl, rl, b, gnd);
        if(run(c, "switch open")) expect("switch open leakage", v(b) / 1000.0, 1.0 / (1e6 + 1000.0), 5e-10);
        pm::variant on{};
        on.boolean = true;
        on.type = pm::variant_type::boolean;
        if(!sw->ptr->set_attribute(0, on)) ++failures;
        if(run(c, "switch closed")) expect("switch closed v", v(b), 1.0, 1e-12);
    }
    {   // generators at t = 0 (DC uses the waveform value at t = 0)
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto& gnd{nl.ground_node};
        constexpr double Vh = 5.0, Vl = 1.0, f = 1000.0, pi = std::numbers::pi;
        pm::model_base* g[8] = {
            add_model(nl, pm::square_gen{.Vh = Vh, .Vl = Vl, .freq = f, .duty = 0.25, .phase = 0.0}).mod,
            add_model(nl, pm::square_gen{.Vh = Vh, .Vl = Vl, .freq = f, .duty = 0.25, .phase = pi}).mod,
            add_model(nl, pm::sawtooth_gen{.Vh = Vh, .Vl = Vl, .freq = f, .phase = 0.0}).mod,
            add_model(nl, pm::sawtooth_gen{.Vh = Vh, .Vl = Vl, .freq = f, .phase = pi}).mod,
            add_model(nl, pm::triangle_gen{.Vh = Vh, .Vl = Vl, .freq = f, .phase = 0.0}).mod,
            add_model(nl, pm::triangle_gen{.Vh = Vh, .Vl = Vl, .freq = f, .phase = pi}).mod,
            add_model(nl, pm::pulse_gen{.Vh = Vh, .Vl = Vl, .freq = f, .duty = 0.1, .phase = 0.0, .tr = 0.0, .tf = 0.0}).mod,
            add_model(nl, pm::pulse_gen{.Vh = Vh, .Vl = Vl, .freq = f, .duty = 0.1, .phase = 1.5 * pi, .tr = 0.0, .tf = 0.0}).mod};
        double const want[8] = {Vh, Vl, Vl, Vl + (Vh - Vl) * 0.5, Vl, Vh, Vh, Vl};
        pm::node_t* n[8];
        for(int i = 0; i < 8; ++i)
        {
            n[i] = &create_node(nl);
            wire2(nl, g[i], *n[i], gnd);
        }
        if(run(c, "generators"))
            for(int i = 0; i < 8; ++i) expect("generator value at t=0", v(*n[i]), want[i], 1e-9);
    }
    {   // coupled inductors, k = 0, 10 steps of 1e-5: primary current rises like an R-L, the secondary stays at rest
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::TR);
        c.get_analyze_setting().tr.t_step = 1e-5;
        c.get_analyze_setting().tr.t_stop = 1e-4;
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 1.0})};
        auto [r1, p1]{add_model(nl, pm::resistance{.r = 10.0})};
        auto [kl, p2]{add_model(nl, pm::coupled_inductors{.L1 = 1e-3, .L2 = 1e-3, .k = 0.0})};
        auto [r2, p3]{add_model(nl, pm::resistance{.r = 10.0})};
        auto& np{create_node(nl)};
        auto& nr{create_node(nl)};
        auto& nsr{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, np, gnd);
        wire4(nl, kl, np, nr, gnd, nsr);
        wire2(nl, r1, nr, gnd);
        wire2(nl, r2, nsr, gnd);
        if(run(c, "coupled inductors"))
        {
            auto const bv{kl->ptr->generate_branch_view()};
            double const i1 = bv.branches[0].current.real(), i2 = bv.branches[1].current.real();
            // trapezoidal R-L: i[n+1] = ((1 - a) i[n] + 2 a V / R) / (1 + a), a = R dt / (2 L); the first step starts from i = 0, v_L(0) = 0
            if(!(i1 > 0.04 && i1 < 0.095)) expect("coupled primary current in (0.04, 0.095)", i1, 0.0632, 0.03);
            expect("coupled secondary current", i2, 0.0, 1e-9);
        }
    }
    {   // IAC drives nothing in DC (IAC.h:124-128)
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::IAC{.m_Ip = 1e-3, .m_omega = 6283.0, .m_phase = 0.5})};
        auto [rl, p1]{add_model(nl, pm::resistance{.r = 1000.0})};
        auto& n1{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, gnd, n1);
        wire2(nl, rl, n1, gnd);
        if(run(c, "iac dc")) expect("iac dc", v(n1), 0.0, 0.0);
    }
    {   // center-tap transformer, n_total = 2: +-1 V on the half windings (test/0005.models/transformer_center_tap_ratio.cpp)
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 4.0})};
        auto [tx, p1]{add_model(nl, pm::transformer_center_tap{.n_total = 2.0})};
        auto [r1, p2]{add_model(nl, pm::resistance{.r = 100.0})};
        auto [r2, p3]{add_model(nl, pm::resistance{.r = 100.0})};
        auto& np{create_node(nl)};
        auto& s1{create_node(nl)};
        auto& s2{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, np, gnd);
        add_to_node(nl, *tx, 0, np);
        add_to_node(nl, *tx, 1, gnd);
        add_to_node(nl, *tx, 2, s1);
        add_to_node(nl, *tx, 3, gnd);
        add_to_node(nl, *tx, 4, s2);
        wire2(nl, r1, s1, gnd);
        wire2(nl, r2, s2, gnd);
        if(run(c, "center tap"))
        {
            expect("center tap vs1", v(s1), 1.0, 1e-9);
            expect("center tap vs2", v(s2), -1.0, 1e-9);
            expect("center tap branches", static_cast<double>(tx->ptr->generate_branch_view().size), 3.0, 0.0);
        }
    }
    {   // relay hysteresis (test/0005.models/relay_hysteresis.cpp): 0 V open, 6 V closes, 4 V stays closed, 2 V opens
        circ c{};
        c.set_analyze_type(::phy_engine::analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto [src, p0]{add_model(nl, pm::VDC{.V = 1.0})};
        auto [ctl, p1]{add_model(nl, pm::VDC{.V = 0.0})};
        auto [rl, p2]{add_model(nl, pm::resistance{.r = 100.0})};
        pm::relay r0{};
        r0.Von = 5.0;
        r0.Voff = 3.0;
        auto [ry, p3]{add_model(nl, r0)};
        auto& a{create_node(nl)};
        auto& b{create_node(nl)};
        auto& nc{create_node(nl)};
        auto& gnd{nl.ground_node};
        wire2(nl, src, a, gnd);
        wire2(nl, rl, b, gnd);
        wire2(nl, ctl, nc, gnd);
        wire4(nl, ry, nc, gnd, a, b);
        double const drive[4] = {0.0, 6.0, 4.0, 2.0};
        double const want[4] = {0.0, 1.0, 1.0, 0.0};
        for(int i = 0; i < 4; ++i)
        {
            pm::variant vv{};
            vv.d = drive[i];
            vv.type = pm::variant_type::d;
            if(!ctl->ptr->set_attribute(0, vv)) ++failures;
            if(run(c, "relay")) expect("relay contact voltage", v(b), want[i], 1e-6);
        }
    }
    if(failures) std::fprintf(stderr, "linear_models: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
