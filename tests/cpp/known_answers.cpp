// tests/cpp/known_answers.cpp -- three circuits with answers known in closed form, built through the plug-in API
// (netlist / add_model / add_to_node / circult::analyze) and solved by the MI355X engine.  exit 0 = pass.
//
//   1. series string on a DC source   v_k = V * (sum of the resistances below tap k) / (sum of all), I = V / sum
//                                     (the 10 ohm + 20 ohm string on 3 V of the reference's test/0004.solver/dc.cpp: 3 V, 2 V, 0.1 A)
//   2. RC charging, trapezoidal rule  after n steps of h:  v_n = V (1 - rho^n) with rho = (1 - h/2RC) / (1 + h/2RC) once the
//                                     companion history is primed; |v(tau) - V (1 - 1/e)| <= 5e-3 is the bound the reference's
//                                     test/0005.models/rc_step_tr.cpp accepts
//   3. diode operating point          the solution satisfies KCL with the Shockley law to Newton's stop tolerance, lies in the
//                                     window 0.5 .. 0.9 V of the reference's test/0011.nonlinear/op_pn_junction.cpp and on the
//                                     value the real reference computes (tests/golden/diode_op.bin: 0.62944165 V)
#include <cmath>
#include <cstdio>
#include <initializer_list>
#include <vector>

#include <phy_engine/phy_engine.h>

namespace
{
    namespace pm = ::phy_engine::model;
    using node = ::phy_engine::model::node_t;

    struct bench
    {
        ::phy_engine::circult c{};
        ::phy_engine::netlist::netlist& nl{c.get_netlist()};
        node& gnd{::phy_engine::netlist::get_ground_node(nl)};
        node& fresh() { return ::phy_engine::netlist::create_node(nl); }
        // a two-terminal element between a and b; returns the model handle
        template <class M>
        auto between(M&& m, node& a, node& b)
        {
            auto [ptr, pos]{::phy_engine::netlist::add_model(nl, static_cast<M&&>(m))};
            ::phy_engine::netlist::add_to_node(nl, *ptr, 0, a);
            ::phy_engine::netlist::add_to_node(nl, *ptr, 1, b);
            return ptr;
        }
        bool run(char const* what)
        {
            if(c.analyze()) return true;
            std::fprintf(stderr, "%s: analyze failed: %s\n", what, c.last_error.c_str());
            return false;
        }
    };
    double volts(node const& n) { return n.node_information.an.voltage.real(); }
    bool near(double got, double want, double tol, char const* what)
    {
        if(std::abs(got - want) <= tol) return true;
        std::fprintf(stderr, "%s: got %.15g, expected %.15g (tolerance %.3g)\n", what, got, want, tol);
        return false;
    }

    bool series_string()
    {
        bench b;
        b.c.set_analyze_type(::phy_engine::analyze_type::DC);
        double const V = 3.0;
        std::vector<double> const ohms{10.0, 20.0};
        std::vector<node*> tap{&b.fresh()};
        auto src{b.between(pm::VDC{.V = V}, *tap[0], b.gnd)};
        for(std::size_t k = 0; k < ohms.size(); ++k)
        {
            node& below{k + 1 < ohms.size() ? b.fresh() : b.gnd};
            b.between(pm::resistance{.r = ohms[k]}, *tap[k], below);
            tap.push_back(&below);
        }
        if(!b.run("series string")) return false;
        double total = 0.0;
        for(double r: ohms) total += r;
        double below = total;
        bool ok = true;
        for(std::size_t k = 0; k < ohms.size(); ++k)
        {
            ok = near(volts(*tap[k]), V * below / total, 1e-12, "series string tap") && ok;
            below -= ohms[k];
        }
        double const i{-src->ptr->generate_branch_view().branches[0].current.real()};
        return near(i, V / total, 1e-12, "series string current") && ok;
    }

    bool rc_charging()
    {
        bench b;
        b.c.set_analyze_type(::phy_engine::analyze_type::TR);
        double const V = 1.0, R = 1e3, C = 1e-9, tau = R * C;
        int const n = 100;
        auto& tr{b.c.get_analyze_setting().tr};
        tr.t_step = tau / n;
        tr.t_stop = tau;
        node& drive{b.fresh()};
        node& out{b.fresh()};
        b.between(pm::VDC{.V = V}, drive, b.gnd);
        b.between(pm::resistance{.r = R}, drive, out);
        b.between(pm::capacitor{.m_kZimag = C}, out, b.gnd);
        if(!b.run("rc charging")) return false;
        // the loop bound is accumulated in floating point (circuit.h:242-254): n or n + 1 steps
        if(!(b.c.tr_duration > 0.99 * tau && b.c.tr_duration < 1.02 * tau))
        {
            std::fprintf(stderr, "rc charging: tr_duration %.9g\n", b.c.tr_duration);
            return false;
        }
        return near(volts(out), V * (1.0 - std::exp(-1.0)), 5e-3, "rc charging v(tau)");
    }

    bool diode_operating_point()
    {
        bench b;
        b.c.set_analyze_type(::phy_engine::analyze_type::OP);
        double const V = 1.0, R = 1e3;
        node& drive{b.fresh()};
        node& anode{b.fresh()};
        b.between(pm::VDC{.V = V}, drive, b.gnd);
        b.between(pm::resistance{.r = R}, drive, anode);
        b.between(pm::PN_junction{}, anode, b.gnd);
        if(!b.run("diode operating point")) return false;
        double const vd{volts(anode)};
        if(!(vd > 0.5 && vd < 0.9))
        {
            std::fprintf(stderr, "diode operating point: %.9g outside 0.5 .. 0.9 V\n", vd);
            return false;
        }
        // Shockley law with the model's defaults (Is 1e-14 A, N 1, 27 C): KCL residual at the converged point
        double const vt{1.380649e-23 * (27.0 + 273.15) / 1.602176634e-19};
        double const id{1e-14 * (std::exp(vd / vt) - 1.0)};
        bool const kcl{near((V - vd) / R, id, 2e-6, "diode operating point KCL")};
        return near(vd, 0.62944165, 1e-7, "diode operating point vs the reference's value") && kcl;
    }
}  // namespace

int main()
{
    int failed = 0;
    if(!series_string()) failed |= 1;
    if(!rc_charging()) failed |= 2;
    if(!diode_operating_point()) failed |= 4;
    return failed;
}
