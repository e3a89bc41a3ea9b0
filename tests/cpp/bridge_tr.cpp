// Config C2: VAC 10 V / 50 Hz -> full_bridge_rectifier -> 1 kOhm || 100 uF, g_min = 1e-12, through the
// C++ plug-in API: v+(5 ms) and v+(30 ms) against the values the real reference produces (tests/golden/bridge_c2).
// Also: a user model WITHOUT the gpu_table_define hook (it only has the reference's iterate_dc_define) is stamped on the host and
// added to the device-side system (host-stamp overlay, include/pe_hip.h pe_hip_set_overlay).
#include <cmath>
#include <cstdio>
#include <numbers>

#include <phy_engine/phy_engine.h>

namespace user
{
    struct host_only_resistor
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"HostOnlyR"};
        inline static constexpr ::phy_engine::model::model_device_type device_type{::phy_engine::model::model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"HR"};
        ::phy_engine::model::pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline bool iterate_dc_define(::phy_engine::model::model_reserve_type_t<host_only_resistor>, host_only_resistor const& m, ::phy_engine::MNA::MNA& mna) noexcept
    {
        auto const a{m.pins[0].nodes->node_index}, b{m.pins[1].nodes->node_index};
        mna.G_ref(a, a) += 1e-3;
        mna.G_ref(a, b) -= 1e-3;
        mna.G_ref(b, a) -= 1e-3;
        mna.G_ref(b, b) += 1e-3;
        return true;
    }
    inline ::phy_engine::model::pin_view generate_pin_view_define(::phy_engine::model::model_reserve_type_t<host_only_resistor>, host_only_resistor& m) noexcept { return {m.pins, 2}; }
}  // namespace user

int main()
{
    using namespace ::phy_engine;
    circult c{};
    c.set_analyze_type(analyze_type::TR);
    c.env.g_min = 1e-12;
    c.get_analyze_setting().tr.t_step = 1e-5;
    c.get_analyze_setting().tr.t_stop = 5e-3;
    auto& nl{c.get_netlist()};
    auto [vac, p0]{add_model(nl, model::VAC{.m_Vp = 10.0, .m_omega = 2.0 * std::numbers::pi * 50.0, .m_phase = 0.0})};
    auto [fbr, p1]{add_model(nl, model::full_bridge_rectifier{})};
    auto [r, p2]{add_model(nl, model::resistance{.r = 1000.0})};
    auto [cap, p3]{add_model(nl, model::capacitor{.m_kZimag = 100e-6})};
    auto& a{create_node(nl)};
    auto& b{create_node(nl)};
    auto& plus{create_node(nl)};
    auto& gnd{nl.ground_node};
    add_to_node(nl, *vac, 0, a);
    add_to_node(nl, *vac, 1, b);
    add_to_node(nl, *fbr, 0, a);
    add_to_node(nl, *fbr, 1, b);
    add_to_node(nl, *fbr, 2, plus);
    add_to_node(nl, *fbr, 3, gnd);
    add_to_node(nl, *r, 0, plus);
    add_to_node(nl, *r, 1, gnd);
    add_to_node(nl, *cap, 0, plus);
    add_to_node(nl, *cap, 1, gnd);
    if(!c.analyze())
    {
        std::fprintf(stderr, "bridge: %s\n", c.last_error.c_str());
        return 1;
    }
    double const v5 = plus.node_information.an.voltage.real();
    // the reference's floating-point loop bound (circuit.h:242-254) makes t_stop = 5e-3 / dt = 1e-5 run 501 steps;
    // real reference at step 501 (oracle/_ref/ref_driver): v+ = 8.52738489
    if(c.last_stats.steps != 501 || std::abs(v5 - 8.52738489) > 2e-8)
    {
        std::fprintf(stderr, "bridge: v+(5ms)=%.12g steps=%lld\n", v5, c.last_stats.steps);
        return 2;
    }
    // second analyze() continues from the resident state
    c.get_analyze_setting().tr.t_stop = 25e-3;
    if(!c.analyze()) return 3;
    double const v30 = plus.node_information.an.voltage.real();
    if(!(v30 > 8.0 && v30 < 8.4))
    {
        std::fprintf(stderr, "bridge: v+(30ms)=%.12g\n", v30);
        return 4;
    }
    // a model that only has the reference's host hooks (no gpu_table_define) runs through the host-stamp overlay: here a 1 kOhm
    // "resistor" whose iterate_dc_define stamps its conductance, in series with a built-in 1 kOhm on 2 V -> 1 V at the tap
    circult c2{};
    c2.set_analyze_type(analyze_type::DC);
    auto& nl2{c2.get_netlist()};
    auto [hr, hp]{add_model(nl2, user::host_only_resistor{})};
    auto [r2, r2p]{add_model(nl2, model::resistance{.r = 1000.0})};
    auto [v2, v2p]{add_model(nl2, model::VDC{.V = 2.0})};
    auto& top{create_node(nl2)};
    auto& tap{create_node(nl2)};
    add_to_node(nl2, *v2, 0, top);
    add_to_node(nl2, *v2, 1, nl2.ground_node);
    add_to_node(nl2, *r2, 0, top);
    add_to_node(nl2, *r2, 1, tap);
    add_to_node(nl2, *hr, 0, tap);
    add_to_node(nl2, *hr, 1, nl2.ground_node);
    if(!c2.analyze())
    {
        std::fprintf(stderr, "host-stamped resistor: %s\n", c2.last_error.c_str());
        return 5;
    }
    if(std::abs(tap.node_information.an.voltage.real() - 1.0) > 1e-12) return 6;
    return 0;
}
