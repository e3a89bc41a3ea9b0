// tests/cpp/user_model_overlay.cpp -- plug-in models that have ONLY the reference's host hooks (iterate_*_define,
// step_changed_tr_define; model/model_refs/concept.h) and no device table: the engine evaluates them on the host once per Newton
// iteration and adds their stamps to the device-side system (host-stamp overlay: include/pe_hip.h pe_hip_set_overlay,
// circult::prepare_overlay).  Every check compares with a closed form or with the built-in model of the same physics.
//   1. user_diode      Shockley diode, its own Newton linearisation (companion conductance + current source), no limiting:
//                      OP of 1 V - 1 kOhm - diode must land on the built-in PN_junction's operating point, and on KCL
//   2. user_capacitor  trapezoidal companion kept inside the model (step_changed_tr_define + iterate_tr_define; open in DC):
//                      the RC charging curve must follow the built-in capacitor's step for step
//   3. user_cubic      i = k v^3 (non-linear conductor) against the analytic root of (V - v) / R = k v^3
//   4. user_picky      the cubic conductor with a check_convergence_define hook that vetoes the first iterate that passes the
//                      engine's Newton test (circuit.h:950-963: the models are consulted after the node / branch test): exactly one
//                      extra Newton iteration, same answer
// exit 0 = pass.
#include <cmath>
#include <cstdio>

#include <phy_engine/phy_engine.h>

namespace user
{
    namespace pm = ::phy_engine::model;
    using mna_t = ::phy_engine::MNA::MNA;

    inline double v_of(pm::pin const& p) noexcept { return p.nodes ? p.nodes->node_information.an.voltage.real() : 0.0; }
    // conductance g between the two pins + current source i0 flowing a -> b
    inline void stamp_norton(mna_t& mna, pm::pin const (&pins)[2], double g, double i0) noexcept
    {
        auto const a{pins[0].nodes->node_index}, b{pins[1].nodes->node_index};
        mna.G_ref(a, a) += g;
        mna.G_ref(a, b) -= g;
        mna.G_ref(b, a) -= g;
        mna.G_ref(b, b) += g;
        mna.I_ref(a) -= i0;
        mna.I_ref(b) += i0;
    }

    // ---- 1. Shockley diode
    struct user_diode
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"UserDiode"};
        inline static constexpr pm::model_device_type device_type{pm::model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"UD"};
        double Is{1e-14};
        double Ut{1.380650524e-23 * (27.0 + 273.15) / 1.6021765314e-19};
        pm::pin pins[2]{{{u8"A"}}, {{u8"K"}}};
    };
    inline pm::pin_view generate_pin_view_define(pm::model_reserve_type_t<user_diode>, user_diode& m) noexcept { return {m.pins, 2}; }
    inline bool iterate_dc_define(pm::model_reserve_type_t<user_diode>, user_diode& m, mna_t& mna) noexcept
    {
        double vd{v_of(m.pins[0]) - v_of(m.pins[1])};
        if(vd > 0.8) vd = 0.8;  // crude step limit of the model's own (keeps exp() finite on the first iterate)
        double const e{std::exp(vd / m.Ut)};
        double const id{m.Is * (e - 1.0)}, gd{m.Is * e / m.Ut};
        stamp_norton(mna, m.pins, gd, id - gd * vd);
        return true;
    }

    // ---- 2. capacitor with its own trapezoidal companion (the recurrence of linear/capacitor.h:106-155)
    struct user_capacitor
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"UserCapacitor"};
        inline static constexpr pm::model_device_type device_type{pm::model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"UC"};
        double C{1e-9};
        double geq{}, ieq{};  // companion of the current step
        pm::pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline pm::pin_view generate_pin_view_define(pm::model_reserve_type_t<user_capacitor>, user_capacitor& m) noexcept { return {m.pins, 2}; }
    inline bool iterate_dc_define(pm::model_reserve_type_t<user_capacitor>, user_capacitor&, mna_t&) noexcept { return true; }  // open circuit
    inline bool step_changed_tr_define(pm::model_reserve_type_t<user_capacitor>, user_capacitor& m, double, double now_step) noexcept
    {
        double const v_prev{v_of(m.pins[0]) - v_of(m.pins[1])};
        double const g_new{2.0 * m.C / now_step};
        m.ieq = -(g_new + m.geq) * v_prev - m.ieq;
        m.geq = g_new;
        return true;
    }
    inline bool iterate_tr_define(pm::model_reserve_type_t<user_capacitor>, user_capacitor& m, mna_t& mna, double) noexcept
    {
        stamp_norton(mna, m.pins, m.geq, m.ieq);
        return true;
    }

    // ---- 3. cubic conductor i = k v^3
    struct user_cubic
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"UserCubic"};
        inline static constexpr pm::model_device_type device_type{pm::model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"UQ"};
        double k{1e-3};
        pm::pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline pm::pin_view generate_pin_view_define(pm::model_reserve_type_t<user_cubic>, user_cubic& m) noexcept { return {m.pins, 2}; }
    inline bool iterate_dc_define(pm::model_reserve_type_t<user_cubic>, user_cubic& m, mna_t& mna) noexcept
    {
        double const v{v_of(m.pins[0]) - v_of(m.pins[1])};
        double const g{3.0 * m.k * v * v}, i{m.k * v * v * v};
        stamp_norton(mna, m.pins, g, i - g * v);
        return true;
    }
    // ---- 4. the same conductor, but its own convergence hook vetoes once per analysis
    struct user_picky
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"UserPicky"};
        inline static constexpr pm::model_device_type device_type{pm::model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"UP"};
        double k{1e-3};
        inline static int vetoes_left{1}, asked{};  // (the netlist owns a clone of the model: the test reads these)
        pm::pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline pm::pin_view generate_pin_view_define(pm::model_reserve_type_t<user_picky>, user_picky& m) noexcept { return {m.pins, 2}; }
    inline bool iterate_dc_define(pm::model_reserve_type_t<user_picky>, user_picky& m, mna_t& mna) noexcept
    {
        double const v{v_of(m.pins[0]) - v_of(m.pins[1])};
        double const g{3.0 * m.k * v * v}, i{m.k * v * v * v};
        stamp_norton(mna, m.pins, g, i - g * v);
        return true;
    }
    inline bool check_convergence_define(pm::model_reserve_type_t<user_picky>, user_picky&) noexcept
    {
        ++user_picky::asked;
        if(user_picky::vetoes_left > 0)
        {
            --user_picky::vetoes_left;
            return false;
        }
        return true;
    }
}  // namespace user

namespace
{
    using namespace ::phy_engine;
    using namespace ::phy_engine::model;
    using namespace ::phy_engine::netlist;

    template <class M>
    auto between(::phy_engine::netlist::netlist& nl, M&& m, node_t& a, node_t& b)
    {
        auto [p, pos]{add_model(nl, static_cast<M&&>(m))};
        add_to_node(nl, *p, 0, a);
        add_to_node(nl, *p, 1, b);
        return p;
    }
    double volts(node_t const& n) { return n.node_information.an.voltage.real(); }

    // 1 V - 1 kOhm - diode: user model vs built-in
    template <class D>
    bool diode_op(D&& d, double& vd, char const* what)
    {
        circult c{};
        c.set_analyze_type(analyze_type::OP);
        auto& nl{c.get_netlist()};
        auto& top{create_node(nl)};
        auto& mid{create_node(nl)};
        between(nl, VDC{.V = 1.0}, top, nl.ground_node);
        between(nl, resistance{.r = 1000.0}, top, mid);
        between(nl, static_cast<D&&>(d), mid, nl.ground_node);
        if(!c.analyze())
        {
            std::fprintf(stderr, "%s: %s\n", what, c.last_error.c_str());
            return false;
        }
        vd = volts(mid);
        return true;
    }

    template <class C>
    bool rc_curve(C&& cap, double (&v)[3], char const* what)
    {
        circult c{};
        c.set_analyze_type(analyze_type::TR);
        double const tau{1e3 * 1e-9};
        c.get_analyze_setting().tr.t_step = tau / 50.0;
        auto& nl{c.get_netlist()};
        auto& top{create_node(nl)};
        auto& out{create_node(nl)};
        between(nl, VDC{.V = 1.0}, top, nl.ground_node);
        between(nl, resistance{.r = 1000.0}, top, out);
        between(nl, static_cast<C&&>(cap), out, nl.ground_node);
        for(int k = 0; k < 3; ++k)  // three consecutive analyze() calls of tau / 2 each: the companion state must carry over
        {
            c.get_analyze_setting().tr.t_stop = tau / 2.0;
            if(!c.analyze())
            {
                std::fprintf(stderr, "%s: %s\n", what, c.last_error.c_str());
                return false;
            }
            v[k] = volts(out);
        }
        return true;
    }
}  // namespace

int main()
{
    int failed = 0;
    {
        double v_user{}, v_builtin{};
        if(!diode_op(user::user_diode{}, v_user, "user diode") || !diode_op(PN_junction{}, v_builtin, "built-in diode")) failed |= 1;
        else
        {
            double const ut{1.380650524e-23 * (27.0 + 273.15) / 1.6021765314e-19};
            double const kcl{(1.0 - v_user) / 1000.0 - 1e-14 * (std::exp(v_user / ut) - 1.0)};
            if(std::abs(v_user - v_builtin) > 2e-6 || std::abs(kcl) > 2e-6)
            {
                std::fprintf(stderr, "user diode: vd=%.12g built-in %.12g KCL residual %.3g\n", v_user, v_builtin, kcl);
                failed |= 1;
            }
        }
    }
    {
        double vu[3]{}, vb[3]{};
        if(!rc_curve(user::user_capacitor{.C = 1e-9}, vu, "user capacitor") || !rc_curve(capacitor{.m_kZimag = 1e-9}, vb, "built-in capacitor")) failed |= 2;
        else
            for(int k = 0; k < 3; ++k)
                if(std::abs(vu[k] - vb[k]) > 1e-12 || !(vu[k] > 0.0))
                {
                    std::fprintf(stderr, "user capacitor: segment %d v=%.15g built-in %.15g\n", k, vu[k], vb[k]);
                    failed |= 2;
                }
    }
    {
        // (V - v) / R = k v^3 with V = 2, R = 1 kOhm, k = 1e-3: v^3 + v - 2 = 0 -> v = 1
        circult c{};
        c.set_analyze_type(analyze_type::DC);
        auto& nl{c.get_netlist()};
        auto& top{create_node(nl)};
        auto& mid{create_node(nl)};
        between(nl, VDC{.V = 2.0}, top, nl.ground_node);
        between(nl, resistance{.r = 1000.0}, top, mid);
        between(nl, user::user_cubic{.k = 1e-3}, mid, nl.ground_node);
        mid.node_information.an.voltage = 0.5;  // Newton start away from the flat point v = 0
        if(!c.analyze())
        {
            std::fprintf(stderr, "user cubic: %s\n", c.last_error.c_str());
            failed |= 4;
        }
        else if(std::abs(volts(mid) - 1.0) > 1e-5)
        {
            std::fprintf(stderr, "user cubic: v=%.12g, expected 1\n", volts(mid));
            failed |= 4;
        }
    }
    {
        // the same circuit twice: with the plain cubic conductor and with the one whose check_convergence hook vetoes once
        auto run = [&](auto model, double& v, long long& iters) -> bool
        {
            circult c{};
            c.set_analyze_type(analyze_type::DC);
            auto& nl{c.get_netlist()};
            auto& top{create_node(nl)};
            auto& mid{create_node(nl)};
            between(nl, VDC{.V = 2.0}, top, nl.ground_node);
            between(nl, resistance{.r = 1000.0}, top, mid);
            between(nl, model, mid, nl.ground_node);
            mid.node_information.an.voltage = 0.5;
            if(!c.analyze())
            {
                std::fprintf(stderr, "user picky: %s\n", c.last_error.c_str());
                return false;
            }
            v = volts(mid);
            iters = c.last_stats.newton_iters;
            return true;
        };
        double v_plain{}, v_picky{};
        long long it_plain{}, it_picky{};
        bool ok = run(user::user_cubic{.k = 1e-3}, v_plain, it_plain);
        ok = ok && run(user::user_picky{.k = 1e-3}, v_picky, it_picky);
        int const asked{user::user_picky::asked}, left{user::user_picky::vetoes_left};
        if(!ok) failed |= 8;
        else if(it_picky != it_plain + 1 || asked != 2 || left != 0 || std::abs(v_picky - 1.0) > 1e-5)
        {
            std::fprintf(stderr, "user picky: %lld iterations against %lld without the hook (expected one more), hook asked %d times, v=%.12g\n", it_picky, it_plain, asked,
                         v_picky);
            failed |= 8;
        }
    }
    return failed;
}
