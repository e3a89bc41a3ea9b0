// models_builtin.h -- the device models of the resident path, with the reference's struct / member names so that
// netlist-building code is source compatible, and the additive gpu_table_define hook instead of host stamping.
//   resistance / capacitor / inductor / VDC / VAC / IDC      model/models/linear/*.h
//   PN_junction / full_bridge_rectifier / nmosfet / pmosfet / BJT_NPN / BJT_PNP   model/models/non-linear/*.h
//   IAC / VCCS / VCVS / CCCS / CCVS / op_amp / transformer / coupled_inductors   model/models/linear/*.h
//   single_pole_switch                                       model/models/controller/switch.h
//   sawtooth_gen / square_gen / pulse_gen / triangle_gen     model/models/generator/*.h
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
    // The reference's host stamp (linear/resistance.h:82-110): the conductance 1/r on the four G cells of its two nodes.  The
    // engine itself stamps built-in models from their device tables; the hook is part of the model's interface
    // (defines::can_iterate_dc) and is what a wrapper model without a table of its own would forward to.
    inline bool iterate_dc_define(model_reserve_type_t<resistance>, resistance const& m, ::phy_engine::MNA::MNA& mna) noexcept
    {
        auto const* a = m.pins[0].nodes;
        auto const* b = m.pins[1].nodes;
        if(!a || !b) return true;
        double const g = 1.0 / m.r;
        mna.G_ref(a->node_index, a->node_index) += g;
        mna.G_ref(a->node_index, b->node_index) -= g;
        mna.G_ref(b->node_index, a->node_index) -= g;
        mna.G_ref(b->node_index, b->node_index) += g;
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
    // (VAC.h:28-84: the attributes are the amplitude, the frequency in Hz and the phase in degrees; the members hold rad/s and rad)
    inline bool set_attribute_define(model_reserve_type_t<VAC>, VAC& m, ::std::size_t n, variant vi) noexcept
    {
        if(n >= 3 || vi.type != variant_type::d) return false;
        if(n == 0) m.m_Vp = vi.d;
        else if(n == 1)
            m.m_omega = vi.d * (2.0 * 3.141592653589793238462643383279502884);
        else
            m.m_phase = vi.d * (3.141592653589793238462643383279502884 / 180.0);
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<VAC>, VAC const& m, ::std::size_t n) noexcept
    {
        return n == 0   ? details::dvar(m.m_Vp)
               : n == 1 ? details::dvar(m.m_omega / (2.0 * 3.141592653589793238462643383279502884))
               : n == 2 ? details::dvar(m.m_phase / (3.141592653589793238462643383279502884 / 180.0))
                        : variant{};
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
        double I{0.2};  // (IDC.h:16)
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
        // derived by prepare_foundation_define from the public parameters, which it never modifies (PN_junction.h:296-354)
        double Is_eff{}, Isr_eff{}, Ut{}, Uth{}, Bv_eff{};
        // state of the HOST stamp hooks below (a PN_junction embedded in a host-stamped model, e.g. the body diodes of the
        // reference's BSIM3v3.2 model; a PN_junction of the netlist itself runs on the device and never touches these)
        double Ud_last{}, geq{}, Ieq{}, tr_hist_current{}, tr_prev_g{};
    };
    // PN_junction.h:296-354 with the reference's constants (the device derives the same quantities from the table row:
    // pe::diode_derive, pe_circuit.cpp); the public parameters are never modified
    inline bool prepare_foundation_define(model_reserve_type_t<PN_junction>, PN_junction& m) noexcept
    {
        m.Is_eff = m.Is * m.Area;
        m.Isr_eff = m.Isr * m.Area;
        m.Ut = 1.380650524e-23 * (m.Temp + 273.15) / 1.6021765314e-19;
        double const nut = m.N * m.Ut;
        m.Bv_eff = m.Bv_set ? m.Bv - nut * ::std::log(m.Ibv / m.Is_eff) : m.Bv;
        m.Uth = nut * ::std::log(nut / (1.4142135623730950488016887242096981 * m.Is_eff));
        return true;
    }
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

    // ---- host stamp hooks of the junction, for user models that embed one and stamp it themselves (the netlist-level PN_junction is a
    // device-table model).  Same arithmetic as the device evaluation of pe_front.hpp (eval_devices / companion_update), i.e.
    // PN_junction.h:10-16 (limexp), :58-109 (critical-voltage limiting with the breakdown mirror), :358-402, :440-503.
    namespace details
    {
        inline double pn_limexp(double x) noexcept { return x > 50.0 ? ::std::exp(50.0) * (1.0 + (x - 50.0)) : (x < -50.0 ? ::std::exp(-50.0) : ::std::exp(x)); }
        inline double pn_vlimit(PN_junction const& m, double Ud) noexcept
        {
            double const Ute{m.N * m.Ut};
            bool const mirrored{m.Bv_set && Ud < ::std::fmin(0.0, -m.Bv_eff + 10.0 * Ute)};
            double const u0{mirrored ? -(Ud + m.Bv_eff) : Ud}, u1{mirrored ? -(m.Ud_last + m.Bv_eff) : m.Ud_last};
            double uf{u0};
            if(u0 > m.Uth && ::std::fabs(u0 - u1) > 2.0 * Ute)
            {
                if(u1 > 0.0)
                {
                    double const arg{(u0 - u1) / Ute};
                    uf = arg > 0.0 ? u1 + Ute * (2.0 + ::std::log(arg - 2.0)) : u1 - Ute * (2.0 + ::std::log(2.0 - arg));
                }
                else
                    uf = Ute * ::std::log(u0 / Ute);
            }
            else if(u0 < 0.0)
            {
                double const arg{u1 > 0.0 ? -1.0 - u1 : 2.0 * u1 - 1.0};
                if(u0 < arg) uf = arg;
            }
            return mirrored ? -(uf + m.Bv_eff) : uf;
        }
        // conductance g between the pins + current source i0 from pin 0 to pin 1
        template <class V>
        inline void pn_norton(PN_junction const& m, ::phy_engine::MNA::MNA& mna, V g, double i0) noexcept
        {
            auto const a{m.pins[0].nodes->node_index}, b{m.pins[1].nodes->node_index};
            mna.G_ref(a, a) += g;
            mna.G_ref(a, b) -= g;
            mna.G_ref(b, a) -= g;
            mna.G_ref(b, b) += g;
            if(i0 != 0.0)
            {
                mna.I_ref(a) -= i0;
                mna.I_ref(b) += i0;
            }
        }
        inline double pn_vd(PN_junction const& m) noexcept
        { return m.pins[0].nodes->node_information.an.voltage.real() - m.pins[1].nodes->node_information.an.voltage.real(); }
    }  // namespace details
    inline bool iterate_dc_define(model_reserve_type_t<PN_junction>, PN_junction& m, ::phy_engine::MNA::MNA& mna) noexcept
    {
        if(!m.pins[0].nodes || !m.pins[1].nodes) return true;
        double const Ud{details::pn_vlimit(m, details::pn_vd(m))};
        m.Ud_last = Ud;
        double const Ute{m.N * m.Ut}, Uter{m.Nr * m.Ut};
        double Id;
        if(m.Bv_set && Ud < -m.Bv_eff)
        {
            double const e{details::pn_limexp(-(m.Bv_eff + Ud) / Ute)};
            Id = -m.Is_eff * e;
            m.geq = m.Is_eff * e / Ute;
        }
        else
        {
            double e{details::pn_limexp(Ud / Ute)};
            m.geq = m.Is_eff * e / Ute;
            Id = m.Is_eff * (e - 1.0);
            e = details::pn_limexp(Ud / Uter);
            m.geq += m.Isr_eff * e / Uter;
            Id += m.Isr_eff * (e - 1.0);
        }
        m.Ieq = Id - Ud * m.geq;
        details::pn_norton(m, mna, m.geq, m.Ieq);
        return true;
    }
    // small signal: the incremental conductance of the last linearisation + the diffusion capacitance tt geq; no source term
    inline bool iterate_ac_define(model_reserve_type_t<PN_junction>, PN_junction& m, ::phy_engine::MNA::MNA& mna, double omega) noexcept
    {
        if(!m.pins[0].nodes || !m.pins[1].nodes) return true;
        details::pn_norton(m, mna, m.geq, 0.0);
        double const cd{m.tt * m.geq};
        if(omega != 0.0 && m.tt > 0.0 && m.geq > 0.0 && cd > 0.0) details::pn_norton(m, mna, ::std::complex<double>{0.0, cd * omega}, 0.0);
        return true;
    }
    inline bool step_changed_tr_define(model_reserve_type_t<PN_junction>, PN_junction& m, double, double nstep) noexcept
    {
        if(!m.pins[0].nodes || !m.pins[1].nodes) return true;
        m.Ud_last = details::pn_vd(m);
        double const cd{m.tt * m.geq};
        if(!(nstep > 0.0) || !(m.tt > 0.0) || !(m.geq > 0.0) || !(cd > 0.0))
        {
            m.tr_hist_current = 0.0;
            m.tr_prev_g = 0.0;
            return true;
        }
        double const g_new{2.0 * cd / nstep};
        m.tr_hist_current = -(g_new + m.tr_prev_g) * m.Ud_last - m.tr_hist_current;
        m.tr_prev_g = g_new;
        return true;
    }
    inline bool iterate_tr_define(model_reserve_type_t<PN_junction>, PN_junction& m, ::phy_engine::MNA::MNA& mna, double) noexcept
    {
        (void)iterate_dc_define(model_reserve_type<PN_junction>, m, mna);
        if(m.pins[0].nodes && m.pins[1].nodes && m.tr_prev_g != 0.0) details::pn_norton(m, mna, m.tr_prev_g, m.tr_hist_current);
        return true;
    }
    inline bool iterate_trop_define(model_reserve_type_t<PN_junction>, PN_junction& m, ::phy_engine::MNA::MNA& mna) noexcept
    { return iterate_dc_define(model_reserve_type<PN_junction>, m, mna); }
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

    // =================================================================== SURVEY.md 8f rank 1: remaining linear stampers
    namespace details
    {
        // single-double-attribute boilerplate shared by the controlled sources
        template <typename M>
        inline bool set1(double M::*f, M& m, ::std::size_t n, variant vi) noexcept { return n == 0 && set_d(f, m, vi); }
        inline constexpr double two_pi{6.283185307179586476925286766559};
        inline constexpr double deg{0.017453292519943295769236907684886};
    }  // namespace details

    // ------------------------------------------------------------------ IAC (linear/IAC.h): attributes Ip, freq [Hz], phase [deg]
    struct IAC
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"IAC"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"IAC"};
        double m_Ip{0.2};
        double m_omega{50.0};
        double m_phase{0.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<IAC>, IAC& m, ::std::size_t n, variant vi) noexcept
    {
        if(vi.type != variant_type::d || n > 2) return false;
        if(n == 0) m.m_Ip = vi.d;
        else if(n == 1)
            m.m_omega = vi.d * details::two_pi;   // IAC.h:42
        else
            m.m_phase = vi.d * details::deg;      // IAC.h:48
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<IAC>, IAC const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.m_Ip) : n == 1 ? details::dvar(m.m_omega / details::two_pi) : n == 2 ? details::dvar(m.m_phase / details::deg) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<IAC>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[3] = {u8"Ip", u8"freq", u8"phase"};
        return n < 3 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<IAC>, IAC& m) noexcept { return {m.pins, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<IAC>, IAC const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_IAC, 0, 1, -1, {m.m_Ip, m.m_omega, m.m_phase}};
        return true;
    }

    // ------------------------------------------------------------------ the four controlled sources (pins S, T, P, Q):
    // VCCS.h (no branch), VCVS.h (1 branch), CCCS.h (1 branch: the sensing short), CCVS.h (2 branches: output, sense)
    struct VCCS
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"VCCS"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"VCCS"};
        double m_g{1.0};
        pin pins[4]{{{u8"S"}}, {{u8"T"}}, {{u8"P"}}, {{u8"Q"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<VCCS>, VCCS& m, ::std::size_t n, variant vi) noexcept { return details::set1(&VCCS::m_g, m, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<VCCS>, VCCS const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.m_g) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<VCCS>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"G"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<VCCS>, VCCS& m) noexcept { return {m.pins, 4}; }
    inline bool gpu_table_define(model_reserve_type_t<VCCS>, VCCS const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_VCCS, 0, 1, -1, {m.m_g}, 2, 3, -1};
        return true;
    }

    struct VCVS
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"VCVS"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"VCVS"};
        double m_mu{1.0};
        pin pins[4]{{{u8"S"}}, {{u8"T"}}, {{u8"P"}}, {{u8"Q"}}};
        branch branches{};
    };
    inline bool set_attribute_define(model_reserve_type_t<VCVS>, VCVS& m, ::std::size_t n, variant vi) noexcept { return details::set1(&VCVS::m_mu, m, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<VCVS>, VCVS const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.m_mu) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<VCVS>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"Mu"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<VCVS>, VCVS& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<VCVS>, VCVS& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<VCVS>, VCVS const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_VCVS, 0, 1, 0, {m.m_mu}, 2, 3, -1};
        return true;
    }

    struct CCCS
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"CCCS"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"CCCS"};
        double m_alpha{10.0};  // (CCCS.h:14)
        pin pins[4]{{{u8"S"}}, {{u8"T"}}, {{u8"P"}}, {{u8"Q"}}};
        branch branches{};
    };
    inline bool set_attribute_define(model_reserve_type_t<CCCS>, CCCS& m, ::std::size_t n, variant vi) noexcept { return details::set1(&CCCS::m_alpha, m, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<CCCS>, CCCS const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.m_alpha) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<CCCS>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"alpha"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<CCCS>, CCCS& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<CCCS>, CCCS& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<CCCS>, CCCS const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_CCCS, 0, 1, 0, {m.m_alpha}, 2, 3, -1};
        return true;
    }

    struct CCVS
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"CCVS"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"CCVS"};
        double m_r{10.0};
        pin pins[4]{{{u8"S"}}, {{u8"T"}}, {{u8"P"}}, {{u8"Q"}}};
        branch branches[2]{};
    };
    inline bool set_attribute_define(model_reserve_type_t<CCVS>, CCVS& m, ::std::size_t n, variant vi) noexcept { return details::set1(&CCVS::m_r, m, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<CCVS>, CCVS const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.m_r) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<CCVS>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"r"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<CCVS>, CCVS& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<CCVS>, CCVS& m) noexcept { return {m.branches, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<CCVS>, CCVS const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_CCVS, 0, 1, 0, {m.m_r}, 2, 3, 1};
        return true;
    }

    // ------------------------------------------------------------------ op_amp (linear/op_amp.h): pins +, -, OUT+, OUT-
    struct op_amp
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"OpAmp"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"OPAMP"};
        double mu{1e5};
        pin pins[4]{{{u8"+"}}, {{u8"-"}}, {{u8"OUT+"}}, {{u8"OUT-"}}};
        branch branches{};
    };
    inline bool set_attribute_define(model_reserve_type_t<op_amp>, op_amp& m, ::std::size_t n, variant vi) noexcept { return details::set1(&op_amp::mu, m, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<op_amp>, op_amp const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.mu) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<op_amp>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"mu"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<op_amp>, op_amp& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<op_amp>, op_amp& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<op_amp>, op_amp const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_OPAMP, 0, 1, 0, {m.mu}, 2, 3, -1};
        return true;
    }

    // ------------------------------------------------------------------ transformer (linear/transformer.h): pins P, Q, S, T; n = Vp / Vs
    struct transformer
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Transformer"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"TX"};
        double n{1.0};
        pin pins[4]{{{u8"P"}}, {{u8"Q"}}, {{u8"S"}}, {{u8"T"}}};
        branch branches[2]{};
    };
    inline bool set_attribute_define(model_reserve_type_t<transformer>, transformer& m, ::std::size_t n, variant vi) noexcept { return details::set1(&transformer::n, m, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<transformer>, transformer const& m, ::std::size_t n) noexcept { return n == 0 ? details::dvar(m.n) : variant{}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<transformer>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"n"} : ::fast_io::u8string_view{}; }
    inline pin_view generate_pin_view_define(model_reserve_type_t<transformer>, transformer& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<transformer>, transformer& m) noexcept { return {m.branches, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<transformer>, transformer const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_XFMR, 0, 1, 0, {m.n}, 2, 3, 1};
        return true;
    }

    // ------------------------------------------------------------------ coupled_inductors (linear/coupled_inductors.h)
    struct coupled_inductors
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Coupled Inductors"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"K"};
        double L1{1e-3};
        double L2{1e-3};
        double k{0.99};
        pin pins[4]{{{u8"P1"}}, {{u8"P2"}}, {{u8"S1"}}, {{u8"S2"}}};
        branch branches[2]{};
    };
    inline bool set_attribute_define(model_reserve_type_t<coupled_inductors>, coupled_inductors& m, ::std::size_t n, variant vi) noexcept
    {
        double coupled_inductors::* const f[3] = {&coupled_inductors::L1, &coupled_inductors::L2, &coupled_inductors::k};
        return n < 3 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<coupled_inductors>, coupled_inductors const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.L1) : n == 1 ? details::dvar(m.L2) : n == 2 ? details::dvar(m.k) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<coupled_inductors>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[3] = {u8"L1", u8"L2", u8"k"};
        return n < 3 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<coupled_inductors>, coupled_inductors& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<coupled_inductors>, coupled_inductors& m) noexcept { return {m.branches, 2}; }
    inline bool gpu_table_define(model_reserve_type_t<coupled_inductors>, coupled_inductors const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_COUPLED_L, 0, 1, 0, {m.L1, m.L2, m.k}, 2, 3, 1};
        return true;
    }

    // ------------------------------------------------------------------ single_pole_switch (controller/switch.h)
    struct single_pole_switch
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"switch"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"switch"};
        bool cut_through{};
        pin pins[2]{{{u8"A"}}, {{u8"B"}}};
        branch branches{};
    };
    inline bool set_attribute_define(model_reserve_type_t<single_pole_switch>, single_pole_switch& m, ::std::size_t n, variant vi) noexcept
    {
        if(n != 0 || vi.type != variant_type::boolean) return false;
        m.cut_through = vi.boolean;
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<single_pole_switch>, single_pole_switch const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::bvar(m.cut_through) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<single_pole_switch>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"Cut Through"} : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<single_pole_switch>, single_pole_switch& m) noexcept { return {m.pins, 2}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<single_pole_switch>, single_pole_switch& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<single_pole_switch>, single_pole_switch const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_SWITCH, 0, 1, 0, {m.cut_through ? 1.0 : 0.0}};
        return true;
    }

    // ------------------------------------------------------------------ the four waveform generators (generator/*.h)
    struct sawtooth_gen
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Sawtooth Wave Generator"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"SAW"};
        double Vh{5.0};
        double Vl{0.0};
        double freq{1e3};
        double phase{0.0};  // radians
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
        branch branches{};
    };
    struct square_gen
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Square Wave Generator"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"SQ"};
        double Vh{5.0};
        double Vl{0.0};
        double freq{1e3};
        double duty{0.5};
        double phase{0.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
        branch branches{};
    };
    struct pulse_gen
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Pulse Wave Generator"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"PULSE"};
        double Vh{5.0};
        double Vl{0.0};
        double freq{1e3};
        double duty{0.5};
        double phase{0.0};
        double tr{0.0};
        double tf{0.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
        branch branches{};
    };
    struct triangle_gen
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Triangle Wave Generator"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"TRIANGLE"};
        double Vh{5.0};
        double Vl{0.0};
        double freq{1e3};
        double phase{0.0};
        pin pins[2]{{{u8"+"}}, {{u8"-"}}};
        branch branches{};
    };
    namespace details
    {
        template <typename G, ::std::size_t N>
        inline bool gen_set(G& g, double G::* const (&f)[N], ::std::size_t n, variant vi) noexcept { return n < N && set_d(f[n], g, vi); }
        template <typename G, ::std::size_t N>
        inline variant gen_get(G const& g, double G::* const (&f)[N], ::std::size_t n) noexcept { return n < N ? dvar(g.*f[n]) : variant{}; }
    }  // namespace details
#define PE_GENERATOR(NAME, TYPE, NATTR, FIELDS, NAMES, ROWPARAMS)                                                                                       \
    inline bool set_attribute_define(model_reserve_type_t<NAME>, NAME& m, ::std::size_t n, variant vi) noexcept                                        \
    {                                                                                                                                                  \
        double NAME::* const f[NATTR] = FIELDS;                                                                                                        \
        return details::gen_set(m, f, n, vi);                                                                                                          \
    }                                                                                                                                                  \
    inline variant get_attribute_define(model_reserve_type_t<NAME>, NAME const& m, ::std::size_t n) noexcept                                          \
    {                                                                                                                                                  \
        double NAME::* const f[NATTR] = FIELDS;                                                                                                        \
        return details::gen_get(m, f, n);                                                                                                              \
    }                                                                                                                                                  \
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<NAME>, ::std::size_t n) noexcept                                   \
    {                                                                                                                                                  \
        constexpr ::fast_io::u8string_view names[NATTR] = NAMES;                                                                                       \
        return n < NATTR ? names[n] : ::fast_io::u8string_view{};                                                                                     \
    }                                                                                                                                                  \
    inline pin_view generate_pin_view_define(model_reserve_type_t<NAME>, NAME& m) noexcept { return {m.pins, 2}; }                                    \
    inline branch_view generate_branch_view_define(model_reserve_type_t<NAME>, NAME& m) noexcept { return {&m.branches, 1}; }                         \
    inline bool gpu_table_define(model_reserve_type_t<NAME>, NAME const& m, gpu_table_rows& t) noexcept                                               \
    {                                                                                                                                                  \
        t.count = 1;                                                                                                                                   \
        t.row[0] = {PE_HIP_VGEN, 0, 1, 0, ROWPARAMS};                                                                                                  \
        return true;                                                                                                                                   \
    }
#define PE_L(...) {__VA_ARGS__}
    // PE_HIP_VGEN columns: type, Vh, Vl, freq, duty, phase, tr, tf
    PE_GENERATOR(sawtooth_gen, 0, 4, PE_L(&sawtooth_gen::Vh, &sawtooth_gen::Vl, &sawtooth_gen::freq, &sawtooth_gen::phase), PE_L(u8"Vh", u8"Vl", u8"freq", u8"phase"),
                 PE_L(0.0, m.Vh, m.Vl, m.freq, 0.5, m.phase, 0.0, 0.0))
    PE_GENERATOR(square_gen, 1, 5, PE_L(&square_gen::Vh, &square_gen::Vl, &square_gen::freq, &square_gen::duty, &square_gen::phase),
                 PE_L(u8"Vh", u8"Vl", u8"freq", u8"duty", u8"phase"), PE_L(1.0, m.Vh, m.Vl, m.freq, m.duty, m.phase, 0.0, 0.0))
    PE_GENERATOR(pulse_gen, 2, 7, PE_L(&pulse_gen::Vh, &pulse_gen::Vl, &pulse_gen::freq, &pulse_gen::duty, &pulse_gen::phase, &pulse_gen::tr, &pulse_gen::tf),
                 PE_L(u8"Vh", u8"Vl", u8"freq", u8"duty", u8"phase", u8"tr", u8"tf"), PE_L(2.0, m.Vh, m.Vl, m.freq, m.duty, m.phase, m.tr, m.tf))
    PE_GENERATOR(triangle_gen, 3, 4, PE_L(&triangle_gen::Vh, &triangle_gen::Vl, &triangle_gen::freq, &triangle_gen::phase), PE_L(u8"Vh", u8"Vl", u8"freq", u8"phase"),
                 PE_L(3.0, m.Vh, m.Vl, m.freq, 0.5, m.phase, 0.0, 0.0))
#undef PE_L
#undef PE_GENERATOR

    // =================================================================== three-pin non-linear devices (re-linearised on the device)
    // Shichman-Hodges level 1 (non-linear/nmosfet.h, pmosfet.h), forward-active Ebers-Moll (non-linear/BJT_NPN.h, BJT_PNP.h)
    struct nmosfet
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"NMOSFET"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"NMOS"};
        double Kp{1e-3};     // A/V^2
        double lambda{0.0};  // 1/V
        double Vth{1.0};     // V
        pin pins[3]{{{u8"D"}}, {{u8"G"}}, {{u8"S"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<nmosfet>, nmosfet& m, ::std::size_t n, variant vi) noexcept
    {
        double nmosfet::* const f[3] = {&nmosfet::Kp, &nmosfet::lambda, &nmosfet::Vth};
        return n < 3 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<nmosfet>, nmosfet const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.Kp) : n == 1 ? details::dvar(m.lambda) : n == 2 ? details::dvar(m.Vth) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<nmosfet>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[3] = {u8"Kp", u8"lambda", u8"Vth"};
        return n < 3 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<nmosfet>, nmosfet& m) noexcept { return {m.pins, 3}; }
    inline bool gpu_table_define(model_reserve_type_t<nmosfet>, nmosfet const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_NMOS, 0, 1, -1, {m.Kp, m.lambda, m.Vth}, 2, -1, -1};
        return true;
    }

    struct pmosfet
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"PMOSFET"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"PMOS"};
        double Kp{1e-3};     // A/V^2
        double lambda{0.0};  // 1/V
        double Vth{1.0};     // V
        pin pins[3]{{{u8"D"}}, {{u8"G"}}, {{u8"S"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<pmosfet>, pmosfet& m, ::std::size_t n, variant vi) noexcept
    {
        double pmosfet::* const f[3] = {&pmosfet::Kp, &pmosfet::lambda, &pmosfet::Vth};
        return n < 3 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<pmosfet>, pmosfet const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.Kp) : n == 1 ? details::dvar(m.lambda) : n == 2 ? details::dvar(m.Vth) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<pmosfet>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[3] = {u8"Kp", u8"lambda", u8"Vth"};
        return n < 3 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<pmosfet>, pmosfet& m) noexcept { return {m.pins, 3}; }
    inline bool gpu_table_define(model_reserve_type_t<pmosfet>, pmosfet const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_PMOS, 0, 1, -1, {m.Kp, m.lambda, m.Vth}, 2, -1, -1};
        return true;
    }

    struct BJT_NPN
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"NPN BJT"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"QNP"};
        double Is{1e-16};
        double N{1.0};
        double BetaF{100.0};
        double Temp{27.0};
        double Area{1.0};
        pin pins[3]{{{u8"B"}}, {{u8"C"}}, {{u8"E"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<BJT_NPN>, BJT_NPN& m, ::std::size_t n, variant vi) noexcept
    {
        double BJT_NPN::* const f[5] = {&BJT_NPN::Is, &BJT_NPN::N, &BJT_NPN::BetaF, &BJT_NPN::Temp, &BJT_NPN::Area};
        return n < 5 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<BJT_NPN>, BJT_NPN const& m, ::std::size_t n) noexcept
    {
        double BJT_NPN::* const f[5] = {&BJT_NPN::Is, &BJT_NPN::N, &BJT_NPN::BetaF, &BJT_NPN::Temp, &BJT_NPN::Area};
        return n < 5 ? details::dvar(m.*f[n]) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<BJT_NPN>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[5] = {u8"Is", u8"N", u8"BetaF", u8"Temp", u8"Area"};
        return n < 5 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<BJT_NPN>, BJT_NPN& m) noexcept { return {m.pins, 3}; }
    inline bool gpu_table_define(model_reserve_type_t<BJT_NPN>, BJT_NPN const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_BJT_NPN, 0, 1, -1, {m.Is, m.N, m.BetaF, m.Temp, m.Area}, 2, -1, -1};
        return true;
    }

    struct BJT_PNP
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"PNP BJT"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"QPN"};
        double Is{1e-16};
        double N{1.0};
        double BetaF{100.0};
        double Temp{27.0};
        double Area{1.0};
        pin pins[3]{{{u8"B"}}, {{u8"C"}}, {{u8"E"}}};
    };
    inline bool set_attribute_define(model_reserve_type_t<BJT_PNP>, BJT_PNP& m, ::std::size_t n, variant vi) noexcept
    {
        double BJT_PNP::* const f[5] = {&BJT_PNP::Is, &BJT_PNP::N, &BJT_PNP::BetaF, &BJT_PNP::Temp, &BJT_PNP::Area};
        return n < 5 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<BJT_PNP>, BJT_PNP const& m, ::std::size_t n) noexcept
    {
        double BJT_PNP::* const f[5] = {&BJT_PNP::Is, &BJT_PNP::N, &BJT_PNP::BetaF, &BJT_PNP::Temp, &BJT_PNP::Area};
        return n < 5 ? details::dvar(m.*f[n]) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<BJT_PNP>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[5] = {u8"Is", u8"N", u8"BetaF", u8"Temp", u8"Area"};
        return n < 5 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<BJT_PNP>, BJT_PNP& m) noexcept { return {m.pins, 3}; }
    inline bool gpu_table_define(model_reserve_type_t<BJT_PNP>, BJT_PNP const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_BJT_PNP, 0, 1, -1, {m.Is, m.N, m.BetaF, m.Temp, m.Area}, 2, -1, -1};
        return true;
    }


    // ------------------------------------------------------------------ relay (controller/relay.h): coil C+ C-, contact A B
    struct relay
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Relay"};
        inline static constexpr model_device_type device_type{model_device_type::non_linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"RELAY"};
        pin pins[4]{{{u8"C+"}}, {{u8"C-"}}, {{u8"A"}}, {{u8"B"}}};
        branch branches{};
        double Von{5.0};
        double Voff{3.0};
        bool engaged{};  // lives on the device (per instance); this member mirrors the reference's layout only
    };
    inline bool set_attribute_define(model_reserve_type_t<relay>, relay& m, ::std::size_t n, variant vi) noexcept
    {
        double relay::* const f[2] = {&relay::Von, &relay::Voff};
        return n < 2 && details::set_d(f[n], m, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<relay>, relay const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.Von) : n == 1 ? details::dvar(m.Voff) : n == 2 ? details::bvar(m.engaged) : variant{};  // (relay.h:51-53: 2 is read only)
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<relay>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[3] = {u8"Von", u8"Voff", u8"Engaged"};
        return n < 3 ? names[n] : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<relay>, relay& m) noexcept { return {m.pins, 4}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<relay>, relay& m) noexcept { return {&m.branches, 1}; }
    inline bool gpu_table_define(model_reserve_type_t<relay>, relay const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_RELAY, 0, 1, 0, {m.Von, m.Voff}, 2, 3, -1};
        return true;
    }

    // ------------------------------------------------------------------ transformer_center_tap (linear/transformer_center_tap.h)
    struct transformer_center_tap
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Transformer Center Tap"};
        inline static constexpr model_device_type device_type{model_device_type::linear};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"TXCT"};
        double n_total{1.0};
        double n_half{};  // 2 * n_total, derived where the stamp is built (pe_circuit.cpp)
        pin pins[5]{{{u8"P"}}, {{u8"Q"}}, {{u8"S1"}}, {{u8"CT"}}, {{u8"S2"}}};
        branch branches[3]{};  // kP, kH1, kH2
    };
    inline bool set_attribute_define(model_reserve_type_t<transformer_center_tap>, transformer_center_tap& m, ::std::size_t n, variant vi) noexcept
    {
        return details::set1(&transformer_center_tap::n_total, m, n, vi);
    }
    inline variant get_attribute_define(model_reserve_type_t<transformer_center_tap>, transformer_center_tap const& m, ::std::size_t n) noexcept
    {
        return n == 0 ? details::dvar(m.n_total) : variant{};
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<transformer_center_tap>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"n_total"} : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<transformer_center_tap>, transformer_center_tap& m) noexcept { return {m.pins, 5}; }
    inline branch_view generate_branch_view_define(model_reserve_type_t<transformer_center_tap>, transformer_center_tap& m) noexcept { return {m.branches, 3}; }
    inline bool gpu_table_define(model_reserve_type_t<transformer_center_tap>, transformer_center_tap const& m, gpu_table_rows& t) noexcept
    {
        t.count = 1;
        t.row[0] = {PE_HIP_XFMR_CT, 0, 1, 0, {m.n_total}, 2, 3, 1, 4, 2};
        return true;
    }
}  // namespace phy_engine::model
