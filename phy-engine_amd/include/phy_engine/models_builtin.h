// models_builtin.h -- the device models of the resident path, with the reference's struct / member names so that
// netlist-building code is source compatible, and the additive gpu_table_define hook instead of host stamping.
//   resistance / capacitor / inductor / VDC / VAC / IDC      model/models/linear/*.h
//   PN_junction / full_bridge_rectifier                      model/models/non-linear/*.h
// The numerics these rows stand for are implemented ONCE, on the device: phy-engine_amd/csrc/pe_front.hpp
// (companion_update / eval_devices), citing the reference lines they follow.
#pragma once
#include "phy_engine_core.h"

namespace phy_engine::model
{
    namespace details
    {
        inline variant dvar(double v) noexcept
        {
            variant r{};
            r.d = v;
            r.type = variant_type::d;
            return r;
        }
        inline variant bvar(bool v) noexcept
        {
            variant r{};
            r.boolean = v;
            r.type = variant_type::boolean;
            return r;
        }
        template <typename M>
        inline bool set_d(double M::*field, M& m, variant vi) noexcept
        {
            if(vi.type != variant_type::d) return false;
            m.*field = vi.d;
            return true;
        }
    }  // namespace details

    // ------------------------------------------------------------------ resistance (linear/resistance.h)
    struct resistance
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Resistance"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"R"};
        double r{10.0};
        pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<resistance>, resistance& m, ::std::size_t n, variant vi) noexcept { return n == 0 && details::set_d(&resistance::r, m, vi); }
    inline variant get_attribute_define(model_reserve_type_t<resistance>, resistance const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.r) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<resistance>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"R"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<resistance>, resistance& m) noexcept { return {m.pins, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<resistance>, resistance const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_R, 0, 1, -1, {m.r}};
        return true;
    }

    // ------------------------------------------------------------------ capacitor (linear/capacitor.h)
    struct capacitor
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Capacitor"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"C"};
        double m_kZimag{1e-5};
        pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<capacitor>, capacitor& m, ::std::size_t n, variant vi) noexcept { return n == 0 && details::set_d(&capacitor::m_kZimag, m, vi); }
    inline variant get_attribute_define(model_reserve_type_t<capacitor>, capacitor const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.m_kZimag) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<capacitor>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"C"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<capacitor>, capacitor& m) noexcept { return {m.pins, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<capacitor>, capacitor const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_C, 0, 1, -1, {m.m_kZimag}};
        return true;
    }

    // ------------------------------------------------------------------ inductor (linear/inductor.h)
    struct inductor
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Inductor"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"I"};
        double m_kZimag{1e-5};
        pin pins[2]{{{u8"A"}}, {{u8"B"}}};
        branch branches{};
    };
    inline bool set_attribute_define(model_reserve_type_t<inductor>, inductor& m, ::std::size_t n, variant vi) noexcept { return n == 0 && details::set_d(&inductor::m_kZimag, m, vi); }
    inline variant get_attribute_define(model_reserve_type_t<inductor>, inductor const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.m_kZimag) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<inductor>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"L"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<inductor>, inductor& m) noexcept { return {m.pins, 2}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<inductor>, inductor& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<inductor>, inductor const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_L, 0, 1, 0, {m.m_kZimag}};
        return true;
    }

    // ------------------------------------------------------------------ VDC (linear/VDC.h; note the member is spelled `branchs`)
    struct VDC
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"VDC"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"VDC"};
        double V{5.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
        branch branchs{};
    };
    inline bool set_attribute_define(model_reserve_type_t<VDC>, VDC& m, ::std::size_t n, variant vi) noexcept { return n == 0 && details::set_d(&VDC::V, m, vi); }
    inline variant get_attribute_define(model_reserve_type_t<VDC>, VDC const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.V) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<VDC>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"V"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<VDC>, VDC& m) noexcept { return {m.pins, 2}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<VDC>, VDC& m) noexcept { return {&m.branchs, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<VDC>, VDC const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_VDC, 0, 1, 0, {m.V}};
        return true;
    }

    // ------------------------------------------------------------------ VAC (linear/VAC.h): Vp, omega [rad/s], phase [rad]
    struct VAC
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"VAC"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"VAC"};
        double m_Vp{5.0};
        double m_omega{50.0};
        double m_phase{0.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
        branch branches{};
    };
    inline bool set_attribute_define(model_reserve_type_t<VAC>, VAC& m, ::std::size_t n, variant vi) noexcept
    {
        double VAC::* const f[3] = {&VAC::m_Vp, &VAC::m_omega, &VAC::m_phase};
        return n < 3 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<VAC>, VAC const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.m_Vp) : n == 1 ? details::dvar(m.m_omega) : n == 2 ? details::dvar(m.m_phase) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<VAC>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[3] = {u8"Vp", u8"freq", u8"phase"};
        return n < 3 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<VAC>, VAC& m) noexcept { return {m.pins, 2}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<VAC>, VAC& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<VAC>, VAC const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_VAC, 0, 1, 0, {m.m_Vp, m.m_omega, m.m_phase}};
        return true;
    }

    // ------------------------------------------------------------------ IDC (linear/IDC.h)
    struct IDC
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"IDC"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"IDC"};
        double I{1.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<IDC>, IDC& m, ::std::size_t n, variant vi) noexcept { return n == 0 && details::set_d(&IDC::I, m, vi); }
    inline variant get_attribute_define(model_reserve_type_t<IDC>, IDC const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.I) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<IDC>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"I"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<IDC>, IDC& m) noexcept { return {m.pins, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<IDC>, IDC const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_IDC, 0, 1, -1, {m.I}};
        return true;
    }

    // ------------------------------------------------------------------ PN_junction (non-linear/PN_junction.h:19-56)
    struct PN_junction
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"PN Junction"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"PN"};
        double Is{1e-14};
        double N{1.0};
        double Isr{0.0};
        double Nr{2.0};
        double Temp{27.0};
        double Ibv{1e-3};
        double Bv{40.0};
        bool Bv_set{true};
        double Area{1.0};
        double tt{0.0};
        pin pins[2]{{{u8"A"}}, {{u8"B"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<PN_junction>, PN_junction& m, ::std::size_t n, variant vi) noexcept
    {
        double PN_junction::* const f[10] = {&PN_junction::Is, &PN_junction::N, &PN_junction::Isr, &PN_junction::Nr, &PN_junction::Temp, &PN_junction::Ibv,
                                            &PN_junction::Bv, nullptr, &PN_junction::Area, &PN_junction::tt};
        if(n == 7)
        {
            if(vi.type != variant_type::boolean) return false;
            m.Bv_set = vi.boolean;
            return true;
        }
        return n < 10 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<PN_junction>, PN_junction const& m, ::std::size_t n) noexcept
    {
        double const v[10] = {m.Is, m.N, m.Isr, m.Nr, m.Temp, m.Ibv, m.Bv, 0.0, m.Area, m.tt};
        if(n == 7) return details::bvar(m.Bv_set);
        return n < 10 ? details::dvar(v[n]) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<PN_junction>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[10] = {u8"Is", u8"N", u8"Isr", u8"Nr", u8"Temp", u8"Ibv", u8"Bv", u8"Bv_set", u8"Area", u8"tt"};
        return n < 10 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<PN_junction>, PN_junction& m) noexcept { return {m.pins, 2}; }
    inline gpu_table_row pn_row(PN_junction const& m, int pa, int pb, bool tt_in_tr) noexcept
    {
        return {PE_HIP_DIODE, pa, pb, -1, {m.Is, m.N, m.Isr, m.Nr, m.Temp, m.Ibv, m.Bv, m.Bv_set ? 1.0 : 0.0, m.Area, m.tt, tt_in_tr ? 1.0 : 0.0}};
    }
    inline bool gpu_table_define(model_reserve_type_t<PN_junction>, PN_junction const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = pn_row(m, 0, 1, true);
        return true;
    }

    // ------------------------------------------------------------------ full_bridge_rectifier (non-linear/full_bridge_rectifier.h:9-24)
    struct full_bridge_rectifier
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Full Bridge Rectifier"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"FBR"};
        pin pins[4]{{{u8"A"}}, {{u8"B"}}, {{u8"+"}}, {{u8"-"}}};
        PN_junction D1{}, D2{}, D3{}, D4{};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<full_bridge_rectifier>, full_bridge_rectifier& m) noexcept { return {m.pins, 4}; }
    inline bool gpu_table_define(model_reserve_type_t<full_bridge_rectifier>, full_bridge_rectifier const& m, gpu_table_rows& t) noexcept
    {
        // D1 A->+, D2 B->+, D3 - -> A, D4 - -> B; no iterate_tr in the reference => the diffusion-cap companion is never stamped
        t.count = 4;
        t.row[0] = pn_row(m.D1, 0, 2, false);
        t.row[1] = pn_row(m.D2, 1, 2, false);
        t.row[2] = pn_row(m.D3, 3, 0, false);
        t.row[3] = pn_row(m.D4, 3, 1, false);
        return true;
    }

    static_assert(model<resistance> && model<capacitor> && model<inductor> && model<VDC> && model<VAC> && model<IDC> && model<PN_junction> && model<full_bridge_rectifier>);
    static_assert(defines::can_gpu_table<resistance> && defines::can_generate_branch_view<VDC> && !defines::can_generate_branch_view<resistance>);
}  // namespace phy_engine::model
