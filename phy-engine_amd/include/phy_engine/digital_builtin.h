// digital_builtin.h -- the digital / mixed-signal primitives config C4 needs, host side, bit exact:
//   comparator   model/models/controller/comparator.h:72-108   (digital device with analog inputs)
//   NOT / AND / OR / OUTPUT / INPUT   model/models/digital/logical/{not,and,or,output,input}.h
// Event logic is integer / enum work on a handful of nodes: it stays on the host by design.  Struct and member names
// follow the reference so netlist-building code is source compatible.
#pragma once
#include "phy_engine_core.h"

namespace phy_engine::model
{
    namespace details
    {
        // Threshold state machine every gate runs on a pin that sits on an analog / hybrid node
        // (logical/not.h:160-251): L -(v >= Hl)-> X pending H for Tsu; H -(v <= Ll)-> X pending L for Th; a pending
        // transition resolves once it has been held long enough, or falls back when the voltage leaves the band.
        struct analog_input_t
        {
            digital_node_statement_t value{digital_node_statement_t::X};
            digital_node_statement_t pending{digital_node_statement_t::X};
            double since{};
        };
        inline void sample_analog_pin(analog_input_t& in, double v, double Ll, double Hl, double Tsu, double Th, double now) noexcept
        {
            using s = digital_node_statement_t;
            switch(in.value)
            {
                case s::false_state:
                    if(v >= Hl)
                    {
                        if(Tsu > 0.0)
                        {
                            in.value = s::indeterminate_state;
                            in.pending = s::true_state;
                            in.since = now;
                        }
                        else
                            in.value = s::true_state;
                    }
                    break;
                case s::true_state:
                    if(v <= Ll)
                    {
                        if(Th > 0.0)
                        {
                            in.value = s::indeterminate_state;
                            in.pending = s::false_state;
                            in.since = now;
                        }
                        else
                            in.value = s::false_state;
                    }
                    break;
                case s::indeterminate_state:
                    if(in.pending == s::false_state)
                    {
                        if(v <= Ll)
                        {
                            if(now - in.since >= Tsu) in.value = s::false_state;
                        }
                        else
                            in.value = s::true_state;
                    }
                    else if(in.pending == s::true_state)
                    {
                        if(v >= Hl)
                        {
                            if(now - in.since >= Th) in.value = s::true_state;
                        }
                        else
                            in.value = s::false_state;
                    }
                    else
                    {
                        if(v >= Hl)
                        {
                            if(now - in.since >= Th) in.value = s::true_state;
                        }
                        else if(v <= Ll)
                        {
                            if(now - in.since >= Tsu) in.value = s::false_state;
                        }
                        else
                            in.since = now;
                    }
                    break;
                default: break;
            }
        }
        template <typename G>
        inline digital_node_statement_t read_input(G& g, analog_input_t& in, node_t* n, double now) noexcept
        {
            if(n->num_of_analog_node != 0) sample_analog_pin(in, n->node_information.an.voltage.real(), g.Ll, g.Hl, g.Tsu, g.Th, now);
            else
                in.value = n->node_information.dn.state;
            return in.value;
        }
        // drive the output node: an analog node becomes an ideal source of the next analyze(); a digital node is
        // written and queued only when the value differs from what this gate produced last
        template <typename G>
        inline ::phy_engine::digital::need_operate_analog_node_t drive_output(G& g, node_t* o, digital_node_statement_t res,
                                                                              ::phy_engine::digital::digital_node_update_table& table) noexcept
        {
            using s = digital_node_statement_t;
            bool const changed = g.last_outputA != res;
            g.last_outputA = res;
            if(o->num_of_analog_node != 0)
            {
                if(res == s::true_state) return {g.Hl, o};
                if(res == s::high_impedence_state) return {};
                return {g.Ll, o};
            }
            o->node_information.dn.state = res;
            if(changed) table.tables.insert(o);
            return {};
        }
    }  // namespace details

    // ------------------------------------------------------------------ comparator (controller/comparator.h)
    struct comparator
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"Comparator"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"CMP"};
        pin pins[3]{{{u8"A"}}, {{u8"B"}}, {{u8"o"}}};
        double Ll{0.0};
        double Hl{5.0};
    };
    inline bool set_attribute_define(model_reserve_type_t<comparator>, comparator& c, ::std::size_t n, variant vi) noexcept
    {
        if(vi.type != variant_type::d || n > 1) return false;
        (n == 0 ? c.Ll : c.Hl) = vi.d;
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<comparator>, comparator const& c, ::std::size_t n) noexcept
    {
        variant r{};
        if(n > 1) return r;
        r.d = n == 0 ? c.Ll : c.Hl;
        r.type = variant_type::d;
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<comparator>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"Ll"} : n == 1 ? ::fast_io::u8string_view{u8"Hl"} : ::fast_io::u8string_view{};
    }
    inline pin_view generate_pin_view_define(model_reserve_type_t<comparator>, comparator& c) noexcept { return {c.pins, 3}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<comparator>, comparator& c,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        auto* a = c.pins[0].nodes;
        auto* b = c.pins[1].nodes;
        auto* o = c.pins[2].nodes;
        if(!a || !b || !o) return {};
        bool const hi = a->node_information.an.voltage.real() >= b->node_information.an.voltage.real();
        if(o->num_of_analog_node != 0) return {hi ? c.Hl : c.Ll, o};
        auto const next = hi ? digital_node_statement_t::true_state : digital_node_statement_t::false_state;
        if(o->node_information.dn.state != next)
        {
            o->node_information.dn.state = next;
            table.tables.insert(o);
        }
        return {};
    }

    // ------------------------------------------------------------------ gates
    struct NOT
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"NOT"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"NOT"};
        pin pins[2]{{{u8"i"}}, {{u8"o"}}};
        double Ll{0.0}, Hl{5.0}, Tsu{1e-9}, Th{5e-10};
        details::analog_input_t inA{};
        digital_node_statement_t last_outputA{digital_node_statement_t::X};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<NOT>, NOT& g) noexcept { return {g.pins, 2}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<NOT>, NOT& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double now,
                                                                                       digital_update_method_t) noexcept
    {
        auto* i = g.pins[0].nodes;
        auto* o = g.pins[1].nodes;
        if(!i || !o) return {};
        return details::drive_output(g, o, ~details::read_input(g, g.inA, i, now), table);
    }

    // two-input gates (digital/logical/and.h, or.h, xor.h, xnor.h, nand.h, nor.h): same input sampling, output = op(a, b)
    template <int OP>  // 0 AND, 1 OR, 2 XOR, 3 XNOR, 4 NAND, 5 NOR
    struct gate2
    {
        inline static constexpr ::fast_io::u8string_view names[6] = {u8"AND", u8"OR", u8"XOR", u8"XNOR", u8"NAND", u8"NOR"};
        inline static constexpr ::fast_io::u8string_view model_name{names[OP]};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{model_name};
        pin pins[3]{{{u8"ia"}}, {{u8"ib"}}, {{u8"o"}}};
        double Ll{0.0}, Hl{5.0}, Tsu{1e-9}, Th{5e-10};
        details::analog_input_t inA{}, inB{};
        digital_node_statement_t last_outputA{digital_node_statement_t::X};
    };
    using AND = gate2<0>;
    using OR = gate2<1>;
    using XOR = gate2<2>;
    using XNOR = gate2<3>;
    using NAND = gate2<4>;
    using NOR = gate2<5>;
    template <int OP>
    inline pin_view generate_pin_view_define(model_reserve_type_t<gate2<OP>>, gate2<OP>& g) noexcept { return {g.pins, 3}; }
    template <int OP>
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<gate2<OP>>, gate2<OP>& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double now,
                                                                                       digital_update_method_t) noexcept
    {
        auto* a = g.pins[0].nodes;
        auto* b = g.pins[1].nodes;
        auto* o = g.pins[2].nodes;
        if(!a || !b || !o) return {};
        auto const va = details::read_input(g, g.inA, a, now);
        auto const vb = details::read_input(g, g.inB, b, now);
        digital_node_statement_t r{};
        if constexpr(OP == 0) r = va & vb;
        else if constexpr(OP == 1)
            r = va | vb;
        else if constexpr(OP == 2)
            r = va ^ vb;
        else if constexpr(OP == 3)
            r = ~(va ^ vb);
        else if constexpr(OP == 4)
            r = ~(va & vb);
        else
            r = ~(va | vb);
        return details::drive_output(g, o, r, table);
    }

    // ------------------------------------------------------------------ YES buffer (digital/logical/yes.h)
    struct YES
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"YES"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"YES"};
        pin pins[2]{{{u8"i"}}, {{u8"o"}}};
        double Ll{0.0}, Hl{5.0}, Tsu{1e-9}, Th{5e-10};
        details::analog_input_t inA{};
        digital_node_statement_t last_outputA{digital_node_statement_t::X};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<YES>, YES& g) noexcept { return {g.pins, 2}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<YES>, YES& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double now,
                                                                                       digital_update_method_t) noexcept
    {
        auto* i = g.pins[0].nodes;
        auto* o = g.pins[1].nodes;
        if(!i || !o) return {};
        return details::drive_output(g, o, details::read_input(g, g.inA, i, now), table);
    }

    // ------------------------------------------------------------------ OUTPUT probe (digital/logical/output.h): attribute 0 = value
    struct OUTPUT
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"OUTPUT"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"OUTPUT"};
        pin pins{{u8"i"}};
        double Ll{0.0}, Hl{5.0}, Tsu{1e-9}, Th{5e-10};
        details::analog_input_t inA{};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<OUTPUT>, OUTPUT& g) noexcept { return {&g.pins, 1}; }
    inline variant get_attribute_define(model_reserve_type_t<OUTPUT>, OUTPUT const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n != 0) return r;
        r.digital = g.inA.value;
        r.type = variant_type::digital;
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<OUTPUT>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"value"} : ::fast_io::u8string_view{}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<OUTPUT>, OUTPUT& g,
                                                                                       ::phy_engine::digital::digital_node_update_table&, double now,
                                                                                       digital_update_method_t) noexcept
    {
        if(auto* i = g.pins.nodes) (void)details::read_input(g, g.inA, i, now);
        return {};
    }

    // ------------------------------------------------------------------ INPUT (digital/logical/input.h:90-135): drives attribute 0 onto its node
    struct INPUT
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"INPUT"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::before_all_clk};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"INPUT"};
        pin pins{{u8"o"}};
        double Ll{0.0}, Hl{5.0};
        digital_node_statement_t outputA{digital_node_statement_t::X};
        digital_node_statement_t last_outputA{digital_node_statement_t::X};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<INPUT>, INPUT& g) noexcept { return {&g.pins, 1}; }
    inline bool set_attribute_define(model_reserve_type_t<INPUT>, INPUT& g, ::std::size_t n, variant vi) noexcept
    {
        if(n != 0 || vi.type != variant_type::digital) return false;
        g.outputA = vi.digital;
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<INPUT>, INPUT const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n != 0) return r;
        r.digital = g.outputA;
        r.type = variant_type::digital;
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<INPUT>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"value"} : ::fast_io::u8string_view{}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<INPUT>, INPUT& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        auto* o = g.pins.nodes;
        if(!o) return {};
        return details::drive_output(g, o, g.outputA, table);
    }

    static_assert(defines::is_valid_digital_model<comparator> && defines::is_valid_digital_model<NOT> && defines::is_valid_digital_model<AND> &&
                  defines::is_valid_digital_model<OUTPUT> && defines::is_valid_digital_model<INPUT>);
}  // namespace phy_engine::model
