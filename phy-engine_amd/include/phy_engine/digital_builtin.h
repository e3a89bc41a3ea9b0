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
    // every gate carries the same four attributes (e.g. digital/logical/and.h:38-130): 0 Ll, 1 Hl, 2 Tsu, 3 Th
    namespace details
    {
        template <typename G>
        inline bool set_gate_attribute(G& g, ::std::size_t n, variant vi) noexcept
        {
            if(n >= 4 || vi.type != variant_type::d) return false;
            (n == 0 ? g.Ll : n == 1 ? g.Hl : n == 2 ? g.Tsu : g.Th) = vi.d;
            return true;
        }
        template <typename G>
        inline variant get_gate_attribute(G const& g, ::std::size_t n) noexcept
        {
            variant r{};
            if(n >= 4) return r;
            r.d = n == 0 ? g.Ll : n == 1 ? g.Hl : n == 2 ? g.Tsu : g.Th;
            r.type = variant_type::d;
            return r;
        }
        inline ::fast_io::u8string_view gate_attribute_name(::std::size_t n) noexcept
        {
            constexpr ::fast_io::u8string_view names[4] = {u8"Ll", u8"Hl", u8"Tsu", u8"Th"};
            return n < 4 ? names[n] : ::fast_io::u8string_view{};
        }
    }  // namespace details
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
    inline bool set_attribute_define(model_reserve_type_t<NOT>, NOT& g, ::std::size_t n, variant vi) noexcept { return details::set_gate_attribute(g, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<NOT>, NOT const& g, ::std::size_t n) noexcept { return details::get_gate_attribute(g, n); }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<NOT>, ::std::size_t n) noexcept { return details::gate_attribute_name(n); }
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
    template <int OP>  // 0 AND, 1 OR, 2 XOR, 3 XNOR, 4 NAND, 5 NOR, 6 IMP (~a | b, implication.h:326), 7 NIMP (a & ~b, non_implication.h:326)
    struct gate2
    {
        inline static constexpr ::fast_io::u8string_view names[8] = {u8"AND", u8"OR", u8"XOR", u8"XNOR", u8"NAND", u8"NOR", u8"IMP", u8"NIMP"};
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
    using IMP = gate2<6>;
    using NIMP = gate2<7>;
    template <int OP>
    inline pin_view generate_pin_view_define(model_reserve_type_t<gate2<OP>>, gate2<OP>& g) noexcept { return {g.pins, 3}; }
    template <int OP>
    inline bool set_attribute_define(model_reserve_type_t<gate2<OP>>, gate2<OP>& g, ::std::size_t n, variant vi) noexcept { return details::set_gate_attribute(g, n, vi); }
    template <int OP>
    inline variant get_attribute_define(model_reserve_type_t<gate2<OP>>, gate2<OP> const& g, ::std::size_t n) noexcept { return details::get_gate_attribute(g, n); }
    template <int OP>
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<gate2<OP>>, ::std::size_t n) noexcept { return details::gate_attribute_name(n); }
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
        else if constexpr(OP == 5)
            r = ~(va | vb);
        else if constexpr(OP == 6)
            r = ~va | vb;
        else
            r = va & ~vb;
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
    inline bool set_attribute_define(model_reserve_type_t<YES>, YES& g, ::std::size_t n, variant vi) noexcept { return details::set_gate_attribute(g, n, vi); }
    inline variant get_attribute_define(model_reserve_type_t<YES>, YES const& g, ::std::size_t n) noexcept { return details::get_gate_attribute(g, n); }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<YES>, ::std::size_t n) noexcept { return details::gate_attribute_name(n); }
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
        digital_node_statement_t outputA{};  // (input.h:24: value-initialised = L)
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
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<INPUT>, ::std::size_t n) noexcept { return n == 0 ? ::fast_io::u8string_view{u8"boolean"} : ::fast_io::u8string_view{}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<INPUT>, INPUT& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        auto* o = g.pins.nodes;
        if(!o) return {};
        return details::drive_output(g, o, g.outputA, table);
    }


    // ------------------------------------------------------------------ blocks (digital/combinational/*.h, logical/tri_state.h)
    // These read their inputs without the setup / hold state machine of the gates: a digital node's state with Z taken as X,
    // an analog node by plain thresholds (e.g. half_adder.h:41-54); outputs go to digital nodes when they differ from what the
    // block wrote last, and the FIRST output that sits on an analog node becomes the ideal source of the next analyze().
    namespace details
    {
        template <typename G>
        inline digital_node_statement_t read_level(G const& g, node_t* n) noexcept
        {
            using s = digital_node_statement_t;
            if(n->num_of_analog_node == 0)
            {
                auto const st = n->node_information.dn.state;
                return st == s::high_impedence_state ? s::indeterminate_state : st;
            }
            double const v = n->node_information.an.voltage.real();
            if(v >= g.Hl) return s::true_state;
            if(v <= g.Ll) return s::false_state;
            return s::indeterminate_state;
        }
        template <typename G>
        inline ::phy_engine::digital::need_operate_analog_node_t level_of(G const& g, node_t* n, digital_node_statement_t v) noexcept
        {
            return {v == digital_node_statement_t::true_state ? g.Hl : g.Ll, n};
        }
        inline digital_node_statement_t toggled(digital_node_statement_t q) noexcept
        {
            // t_ff.h:61: static_cast<state>(!static_cast<bool>(q)) -- X and Z toggle to L
            return static_cast<digital_node_statement_t>(!static_cast<bool>(q));
        }
        inline bool settled(digital_node_statement_t v) noexcept { return v == digital_node_statement_t::false_state || v == digital_node_statement_t::true_state; }
    }  // namespace details

    // tri-state buffer (logical/tri_state.h:77-131): disabled -> Z on a digital output (queued every time), nothing on an analog one
    struct TRI
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"TRI"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"TRI"};
        pin pins[3]{{{u8"i"}}, {{u8"en"}}, {{u8"o"}}};
        double Ll{0.0}, Hl{5.0};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<TRI>, TRI& t) noexcept { return {t.pins, 3}; }
    inline bool set_attribute_define(model_reserve_type_t<TRI>, TRI& t, ::std::size_t n, variant vi) noexcept
    {
        if(vi.type != variant_type::d || n > 1) return false;
        (n == 0 ? t.Ll : t.Hl) = vi.d;
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<TRI>, TRI const& t, ::std::size_t n) noexcept
    {
        variant r{};
        if(n > 1) return r;
        r.d = n == 0 ? t.Ll : t.Hl;
        r.type = variant_type::d;
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<TRI>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"Ll"} : n == 1 ? ::fast_io::u8string_view{u8"Hl"} : ::fast_io::u8string_view{};
    }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<TRI>, TRI& t,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        auto* i = t.pins[0].nodes;
        auto* en = t.pins[1].nodes;
        auto* o = t.pins[2].nodes;
        if(!i || !en || !o) return {};
        bool const enabled = en->num_of_analog_node != 0 ? en->node_information.an.voltage.real() >= t.Hl : en->node_information.dn.state == s::true_state;
        if(!enabled)
        {
            if(o->num_of_analog_node == 0)
            {
                o->node_information.dn.state = s::high_impedence_state;
                table.tables.insert(o);
            }
            return {};
        }
        s in{};
        if(i->num_of_analog_node != 0)
        {
            double const v = i->node_information.an.voltage.real();
            in = v >= t.Hl ? s::true_state : (v <= t.Ll ? s::false_state : s::indeterminate_state);
        }
        else
            in = i->node_information.dn.state;
        if(o->num_of_analog_node != 0) return details::level_of(t, o, in);
        o->node_information.dn.state = in;
        table.tables.insert(o);
        return {};
    }

    // adders / subtractors / 2 x 2 multiplier: NIN inputs, NOUT outputs, any unknown input makes every output X
    template <int KIND>  // 0 HALF_ADDER, 1 FULL_ADDER, 2 HALF_SUBTRACTOR, 3 FULL_SUBTRACTOR, 4 MUL2
    struct arith_block
    {
        inline static constexpr ::fast_io::u8string_view names[5] = {u8"HALF_ADDER", u8"FULL_ADDER", u8"HALF_SUB", u8"FULL_SUB", u8"MUL2"};
        inline static constexpr ::fast_io::u8string_view ids[5] = {u8"HA", u8"FA", u8"HS", u8"FS", u8"M2"};
        inline static constexpr int n_in = KIND == 0 || KIND == 2 ? 2 : (KIND == 4 ? 4 : 3);
        inline static constexpr int n_out = KIND == 4 ? 4 : 2;
        inline static constexpr ::fast_io::u8string_view model_name{names[KIND]};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{ids[KIND]};
        pin pins[n_in + n_out]{};
        double Ll{0.0}, Hl{5.0};
        digital_node_statement_t last[n_out]{digital_node_statement_t::X, digital_node_statement_t::X};
        constexpr arith_block() noexcept
        {
            constexpr char8_t const* pn[5][8] = {{u8"ia", u8"ib", u8"s", u8"c"},
                                                 {u8"ia", u8"ib", u8"cin", u8"s", u8"cout"},
                                                 {u8"ia", u8"ib", u8"d", u8"b"},
                                                 {u8"ia", u8"ib", u8"bin", u8"d", u8"bout"},
                                                 {u8"a0", u8"a1", u8"b0", u8"b1", u8"p0", u8"p1", u8"p2", u8"p3"}};
            for(int k = 0; k < n_in + n_out; ++k) pins[k].name = ::fast_io::u8string_view{pn[KIND][k]};
            for(int k = 0; k < n_out; ++k) last[k] = digital_node_statement_t::X;
        }
    };
    using HALF_ADDER = arith_block<0>;
    using FULL_ADDER = arith_block<1>;
    using HALF_SUB = arith_block<2>;
    using FULL_SUB = arith_block<3>;
    using MUL2 = arith_block<4>;
    template <int KIND>
    inline pin_view generate_pin_view_define(model_reserve_type_t<arith_block<KIND>>, arith_block<KIND>& g) noexcept
    {
        return {g.pins, static_cast<::std::size_t>(arith_block<KIND>::n_in + arith_block<KIND>::n_out)};
    }
    template <int KIND>
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<arith_block<KIND>>, arith_block<KIND>& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        constexpr int NI = arith_block<KIND>::n_in, NO = arith_block<KIND>::n_out;
        for(int k = 0; k < NI + NO; ++k)
            if(!g.pins[k].nodes) return {};
        s in[4]{};
        bool unknown = false;
        for(int k = 0; k < NI; ++k)
        {
            in[k] = details::read_level(g, g.pins[k].nodes);
            unknown = unknown || in[k] == s::indeterminate_state;
        }
        s out[4]{s::indeterminate_state, s::indeterminate_state, s::indeterminate_state, s::indeterminate_state};
        if(!unknown)
        {
            if constexpr(KIND == 0) out[0] = in[0] ^ in[1], out[1] = in[0] & in[1];                                      // half_adder.h:58-59
            else if constexpr(KIND == 1)
                out[0] = in[0] ^ in[1] ^ in[2], out[1] = (in[0] & in[1]) | (in[0] & in[2]) | (in[1] & in[2]);             // full_adder.h:62-63
            else if constexpr(KIND == 2)
                out[0] = in[0] ^ in[1], out[1] = ~in[0] & in[1];                                                          // half_subtractor.h:72-73
            else if constexpr(KIND == 3)
                out[0] = in[0] ^ in[1] ^ in[2], out[1] = (~in[0] & in[1]) | (~in[0] & in[2]) | (in[1] & in[2]);           // full_subtractor.h:75-76
            else
            {
                // mul2.h:81-88: inputs a0 a1 b0 b1
                auto const t1 = in[0] & in[3], t2 = in[1] & in[2], c1 = t1 & t2, t3 = in[1] & in[3];
                out[0] = in[0] & in[2];
                out[1] = t1 ^ t2;
                out[2] = t3 ^ c1;
                out[3] = t3 & c1;
            }
        }
        if constexpr(KIND == 4)
        {
            // mul2.h:98-121: output by output; the first analog one ends the update
            for(int k = 0; k < NO; ++k)
            {
                auto* n = g.pins[NI + k].nodes;
                if(n->num_of_analog_node != 0) return details::level_of(g, n, out[k]);
                if(g.last[k] != out[k])
                {
                    g.last[k] = out[k];
                    n->node_information.dn.state = out[k];
                    table.tables.insert(n);
                }
            }
            return {};
        }
        else
        {
            // half_adder.h:63-104: digital outputs first, then the first analog one
            for(int k = 0; k < NO; ++k)
            {
                auto* n = g.pins[NI + k].nodes;
                if(n->num_of_analog_node == 0 && g.last[k] != out[k])
                {
                    g.last[k] = out[k];
                    n->node_information.dn.state = out[k];
                    table.tables.insert(n);
                }
            }
            for(int k = 0; k < NO; ++k)
            {
                auto* n = g.pins[NI + k].nodes;
                if(n->num_of_analog_node != 0) return details::level_of(g, n, out[k]);
            }
            return {};
        }
    }

    // edge-triggered flip-flops (d_ff.h, t_ff.h, t_bar_ff.h, jk_ff.h): rising edge = last settled clock level L, now H
    template <int KIND>  // 0 DFF (d clk q), 1 TFF (t clk q), 2 T_BAR_FF (t_bar clk q), 3 JKFF (j k clk q)
    struct flip_flop
    {
        inline static constexpr ::fast_io::u8string_view names[4] = {u8"DFF", u8"TFF", u8"T_BAR_FF", u8"JKFF"};
        inline static constexpr int n_pins = KIND == 3 ? 4 : 3;
        inline static constexpr ::fast_io::u8string_view model_name{names[KIND]};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{names[KIND]};
        pin pins[n_pins]{};
        double Ll{0.0}, Hl{5.0};
        digital_node_statement_t q{digital_node_statement_t::false_state};
        digital_node_statement_t last_clk{digital_node_statement_t::false_state};
        constexpr flip_flop() noexcept
        {
            constexpr char8_t const* pn[4][4] = {{u8"d", u8"clk", u8"q"}, {u8"t", u8"clk", u8"q"}, {u8"t_bar", u8"clk", u8"q"}, {u8"j", u8"k", u8"clk", u8"q"}};
            for(int k = 0; k < n_pins; ++k) pins[k].name = ::fast_io::u8string_view{pn[KIND][k]};
        }
    };
    using DFF = flip_flop<0>;
    using TFF = flip_flop<1>;
    using T_BAR_FF = flip_flop<2>;
    using JKFF = flip_flop<3>;
    template <int KIND>
    inline pin_view generate_pin_view_define(model_reserve_type_t<flip_flop<KIND>>, flip_flop<KIND>& g) noexcept
    {
        return {g.pins, static_cast<::std::size_t>(flip_flop<KIND>::n_pins)};
    }
    template <int KIND>
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<flip_flop<KIND>>, flip_flop<KIND>& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        constexpr int NP = flip_flop<KIND>::n_pins;
        for(int k = 0; k < NP; ++k)
            if(!g.pins[k].nodes) return {};
        auto const a = details::read_level(g, g.pins[0].nodes);
        auto const b = KIND == 3 ? details::read_level(g, g.pins[1].nodes) : s::false_state;
        auto const clk = details::read_level(g, g.pins[NP - 2].nodes);
        auto* nq = g.pins[NP - 1].nodes;
        if(g.last_clk == s::false_state && clk == s::true_state)
        {
            if constexpr(KIND == 0) g.q = a;                                           // d_ff.h:62
            else if constexpr(KIND == 1)
            {
                if(a == s::true_state) g.q = details::toggled(g.q);                    // t_ff.h:59-66
                else if(a == s::indeterminate_state)
                    g.q = s::indeterminate_state;
            }
            else if constexpr(KIND == 2)
            {
                if(a == s::false_state) g.q = details::toggled(g.q);                   // t_bar_ff.h: active-low toggle
                else if(a == s::indeterminate_state)
                    g.q = s::indeterminate_state;
            }
            else
            {
                if(a == s::true_state && b == s::false_state) g.q = s::true_state;     // jk_ff.h:62-79
                else if(a == s::false_state && b == s::true_state)
                    g.q = s::false_state;
                else if(a == s::true_state && b == s::true_state)
                    g.q = details::toggled(g.q);
                else if(a == s::indeterminate_state || b == s::indeterminate_state)
                    g.q = s::indeterminate_state;
            }
        }
        if(details::settled(clk)) g.last_clk = clk;
        if(nq->num_of_analog_node != 0) return details::level_of(g, nq, g.q);
        if(nq->node_information.dn.state != g.q)
        {
            nq->node_information.dn.state = g.q;
            table.tables.insert(nq);
        }
        return {};
    }

    // 4-bit counter (combinational/counter4.h): pins q3 q2 q1 q0 clk en (en missing / Z = enabled); attributes value, unknown
    struct COUNTER4
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"COUNTER4"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"COUNTER4"};
        pin pins[6]{{{u8"q3"}}, {{u8"q2"}}, {{u8"q1"}}, {{u8"q0"}}, {{u8"clk"}}, {{u8"en"}}};
        double Ll{0.0}, Hl{5.0};
        ::std::uint8_t value{};
        bool unknown{};
        digital_node_statement_t last_clk{digital_node_statement_t::false_state};
        ::std::uint8_t last_value{0xFF};
        bool last_unknown{true};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<COUNTER4>, COUNTER4& g) noexcept { return {g.pins, 6}; }
    inline bool set_attribute_define(model_reserve_type_t<COUNTER4>, COUNTER4& g, ::std::size_t n, variant vi) noexcept
    {
        if(n == 0 && vi.type == variant_type::ui8)
        {
            g.value = static_cast<::std::uint8_t>(vi.ui8 & 0x0F);
            g.unknown = false;
            return true;
        }
        if(n == 1 && vi.type == variant_type::boolean)
        {
            g.unknown = vi.boolean;
            return true;
        }
        return false;
    }
    inline variant get_attribute_define(model_reserve_type_t<COUNTER4>, COUNTER4 const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n == 0)
        {
            r.ui8 = static_cast<::std::uint_least8_t>(g.value & 0x0F);
            r.type = variant_type::ui8;
        }
        else if(n == 1)
        {
            r.boolean = g.unknown;
            r.type = variant_type::boolean;
        }
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<COUNTER4>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"value"} : n == 1 ? ::fast_io::u8string_view{u8"unknown"} : ::fast_io::u8string_view{};
    }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<COUNTER4>, COUNTER4& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        auto* n_clk = g.pins[4].nodes;
        auto* n_en = g.pins[5].nodes;
        if(!n_clk) return {};
        auto const clk = details::read_level(g, n_clk);
        // counter4.h:117-119: the enable keeps Z on a digital node (-> enabled); an analog node goes through the thresholds
        s en = !n_en ? s::high_impedence_state : (n_en->num_of_analog_node == 0 ? n_en->node_information.dn.state : details::read_level(g, n_en));
        if(en == s::high_impedence_state) en = s::true_state;
        if(g.last_clk == s::false_state && clk == s::true_state)
        {
            if(en == s::true_state)
            {
                if(!g.unknown) g.value = static_cast<::std::uint8_t>((g.value + 1u) & 0x0F);
            }
            else if(en != s::false_state)
                g.unknown = true;
        }
        if(details::settled(clk)) g.last_clk = clk;
        ::phy_engine::digital::need_operate_analog_node_t drive{};
        for(int pin = 0; pin < 4; ++pin)
        {
            auto* nq = g.pins[pin].nodes;
            if(!nq) continue;
            s const out = g.unknown ? s::indeterminate_state : (((g.value >> static_cast<unsigned>(3 - pin)) & 1u) ? s::true_state : s::false_state);
            if(nq->num_of_analog_node == 0)
            {
                if(nq->node_information.dn.state != out)
                {
                    nq->node_information.dn.state = out;
                    table.tables.insert(nq);
                }
            }
            else if(drive.need_to_operate_analog_node == nullptr)
                drive = details::level_of(g, nq, out);
        }
        g.last_value = g.value;
        g.last_unknown = g.unknown;
        return drive;
    }

    // D latch (combinational/d_latch.h): pins d en q; en X -> q X, en H -> q = d, en L holds; q starts X
    struct DLATCH
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"DLATCH"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"DLATCH"};
        pin pins[3]{{{u8"d"}}, {{u8"en"}}, {{u8"q"}}};
        double Ll{0.0}, Hl{5.0};
        digital_node_statement_t q{digital_node_statement_t::indeterminate_state};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<DLATCH>, DLATCH& g) noexcept { return {g.pins, 3}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<DLATCH>, DLATCH& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        auto* nq = g.pins[2].nodes;
        if(!g.pins[0].nodes || !g.pins[1].nodes || !nq) return {};
        auto const d = details::read_level(g, g.pins[0].nodes);
        auto const en = details::read_level(g, g.pins[1].nodes);
        if(en == s::indeterminate_state) g.q = s::indeterminate_state;
        else if(en == s::true_state)
            g.q = d;
        if(nq->num_of_analog_node != 0) return details::level_of(g, nq, g.q);
        if(nq->node_information.dn.state != g.q)
        {
            nq->node_information.dn.state = g.q;
            table.tables.insert(nq);
        }
        return {};
    }

    // D flip-flop with asynchronous active-low reset (combinational/d_ff_arstn.h): pins d clk arst_n q; reset X -> q X, reset L -> reset_value
    struct DFF_ARSTN
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"DFF_ARSTN"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"DFF_ARSTN"};
        pin pins[4]{{{u8"d"}}, {{u8"clk"}}, {{u8"arst_n"}}, {{u8"q"}}};
        double Ll{0.0}, Hl{5.0};
        digital_node_statement_t q{digital_node_statement_t::false_state};
        digital_node_statement_t reset_value{digital_node_statement_t::false_state};
        digital_node_statement_t last_clk{digital_node_statement_t::false_state};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<DFF_ARSTN>, DFF_ARSTN& g) noexcept { return {g.pins, 4}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<DFF_ARSTN>, DFF_ARSTN& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        for(auto& p: g.pins)
            if(!p.nodes) return {};
        auto const d = details::read_level(g, g.pins[0].nodes);
        auto const clk = details::read_level(g, g.pins[1].nodes);
        auto const rst = details::read_level(g, g.pins[2].nodes);
        auto* nq = g.pins[3].nodes;
        if(rst == s::indeterminate_state) g.q = s::indeterminate_state;
        else if(rst == s::false_state)
            g.q = g.reset_value;
        else if(g.last_clk == s::false_state && clk == s::true_state)
            g.q = d;
        if(details::settled(clk)) g.last_clk = clk;
        if(nq->num_of_analog_node != 0) return details::level_of(g, nq, g.q);
        if(nq->node_information.dn.state != g.q)
        {
            nq->node_information.dn.state = g.q;
            table.tables.insert(nq);
        }
        return {};
    }

    // 4-bit pseudo-random generator (combinational/random_generator4.h): pins q3 q2 q1 q0 clk reset_n; shift register with
    // feedback (b3 ^ b2) ^ 1; reset dominates; attributes state, unknown
    struct RANDOM_GENERATOR4
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"RANDOM_GENERATOR4"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"RANDOM_GENERATOR4"};
        pin pins[6]{{{u8"q3"}}, {{u8"q2"}}, {{u8"q1"}}, {{u8"q0"}}, {{u8"clk"}}, {{u8"reset_n"}}};
        double Ll{0.0}, Hl{5.0};
        ::std::uint8_t state{1u};
        bool unknown{};
        digital_node_statement_t last_clk{digital_node_statement_t::false_state};
        ::std::uint8_t last_state{0xFF};
        bool last_unknown{true};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<RANDOM_GENERATOR4>, RANDOM_GENERATOR4& g) noexcept { return {g.pins, 6}; }
    inline bool set_attribute_define(model_reserve_type_t<RANDOM_GENERATOR4>, RANDOM_GENERATOR4& g, ::std::size_t n, variant vi) noexcept
    {
        if(n == 0 && vi.type == variant_type::ui8)
        {
            g.state = static_cast<::std::uint8_t>(vi.ui8 & 0x0F);
            g.unknown = false;
            return true;
        }
        if(n == 1 && vi.type == variant_type::boolean)
        {
            g.unknown = vi.boolean;
            return true;
        }
        return false;
    }
    inline variant get_attribute_define(model_reserve_type_t<RANDOM_GENERATOR4>, RANDOM_GENERATOR4 const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n == 0)
        {
            r.ui8 = static_cast<::std::uint_least8_t>(g.state & 0x0F);
            r.type = variant_type::ui8;
        }
        else if(n == 1)
        {
            r.boolean = g.unknown;
            r.type = variant_type::boolean;
        }
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<RANDOM_GENERATOR4>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"state"} : n == 1 ? ::fast_io::u8string_view{u8"unknown"} : ::fast_io::u8string_view{};
    }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<RANDOM_GENERATOR4>, RANDOM_GENERATOR4& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        auto* n_clk = g.pins[4].nodes;
        auto* n_rst = g.pins[5].nodes;
        if(!n_clk) return {};
        auto const clk = details::read_level(g, n_clk);
        s rst = !n_rst ? s::high_impedence_state : (n_rst->num_of_analog_node == 0 ? n_rst->node_information.dn.state : details::read_level(g, n_rst));
        if(rst == s::high_impedence_state) rst = s::true_state;
        if(rst == s::false_state)
        {
            g.state = 0u;
            g.unknown = false;
        }
        else if(rst == s::indeterminate_state)
            g.unknown = true;
        else if(g.last_clk == s::false_state && clk == s::true_state)
        {
            if(!g.unknown)
            {
                bool const b3 = ((g.state >> 3u) & 1u) != 0u, b2 = ((g.state >> 2u) & 1u) != 0u;
                bool const feedback = (b3 ^ b2) ^ true;  // random_generator4.h:118-122: the constant keeps it out of the all-zero lock
                g.state = static_cast<::std::uint8_t>(((g.state << 1u) & 0x0E) | static_cast<::std::uint8_t>(feedback));
            }
        }
        if(details::settled(clk)) g.last_clk = clk;
        ::phy_engine::digital::need_operate_analog_node_t drive{};
        for(int pin = 0; pin < 4; ++pin)
        {
            auto* nq = g.pins[pin].nodes;
            if(!nq) continue;
            s const out = g.unknown ? s::indeterminate_state : (((g.state >> static_cast<unsigned>(3 - pin)) & 1u) ? s::true_state : s::false_state);
            if(nq->num_of_analog_node == 0)
            {
                if(nq->node_information.dn.state != out)
                {
                    nq->node_information.dn.state = out;
                    table.tables.insert(nq);
                }
            }
            else if(drive.need_to_operate_analog_node == nullptr)
                drive = details::level_of(g, nq, out);
        }
        g.last_state = g.state;
        g.last_unknown = g.unknown;
        return drive;
    }

    // 8-bit input (logical/eight_bit_input.h): pins b7..b0, drives attribute 0 (value) before every clock; digital pins are
    // rewritten only when the value changed since the last tick
    struct EIGHT_BIT_INPUT
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"EIGHT_BIT_INPUT"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::before_all_clk};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"EIGHT_BIT_INPUT"};
        pin pins[8]{{{u8"b7"}}, {{u8"b6"}}, {{u8"b5"}}, {{u8"b4"}}, {{u8"b3"}}, {{u8"b2"}}, {{u8"b1"}}, {{u8"b0"}}};
        double Ll{0.0}, Hl{5.0};
        ::std::uint8_t value{};
        ::std::uint8_t last_value{0xFF};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<EIGHT_BIT_INPUT>, EIGHT_BIT_INPUT& g) noexcept { return {g.pins, 8}; }
    inline bool set_attribute_define(model_reserve_type_t<EIGHT_BIT_INPUT>, EIGHT_BIT_INPUT& g, ::std::size_t n, variant vi) noexcept
    {
        if(n != 0 || vi.type != variant_type::ui8) return false;
        g.value = static_cast<::std::uint8_t>(vi.ui8);
        return true;
    }
    inline variant get_attribute_define(model_reserve_type_t<EIGHT_BIT_INPUT>, EIGHT_BIT_INPUT const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n != 0) return r;
        r.ui8 = g.value;
        r.type = variant_type::ui8;
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<EIGHT_BIT_INPUT>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"value"} : ::fast_io::u8string_view{};
    }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<EIGHT_BIT_INPUT>, EIGHT_BIT_INPUT& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        bool const changed = g.last_value != g.value;
        ::phy_engine::digital::need_operate_analog_node_t drive{};
        for(int pin = 0; pin < 8; ++pin)
        {
            auto* n = g.pins[pin].nodes;
            if(!n) continue;
            s const out = ((g.value >> static_cast<unsigned>(7 - pin)) & 1u) ? s::true_state : s::false_state;
            if(n->num_of_analog_node == 0)
            {
                if(changed && n->node_information.dn.state != out)
                {
                    n->node_information.dn.state = out;
                    table.tables.insert(n);
                }
            }
            else if(drive.need_to_operate_analog_node == nullptr)
                drive = details::level_of(g, n, out);
        }
        g.last_value = g.value;
        return drive;
    }

    // 8-bit display (logical/eight_bit_display.h): samples b7..b0 into attributes value / unknown_mask (bit set = X or Z)
    struct EIGHT_BIT_DISPLAY
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"EIGHT_BIT_DISPLAY"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"EIGHT_BIT_DISPLAY"};
        pin pins[8]{{{u8"b7"}}, {{u8"b6"}}, {{u8"b5"}}, {{u8"b4"}}, {{u8"b3"}}, {{u8"b2"}}, {{u8"b1"}}, {{u8"b0"}}};
        double Ll{0.0}, Hl{5.0};
        ::std::uint8_t value{};
        ::std::uint8_t unknown_mask{};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<EIGHT_BIT_DISPLAY>, EIGHT_BIT_DISPLAY& g) noexcept { return {g.pins, 8}; }
    inline bool set_attribute_define(model_reserve_type_t<EIGHT_BIT_DISPLAY>, EIGHT_BIT_DISPLAY&, ::std::size_t, variant) noexcept { return false; }
    inline variant get_attribute_define(model_reserve_type_t<EIGHT_BIT_DISPLAY>, EIGHT_BIT_DISPLAY const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n > 1) return r;
        r.ui8 = n == 0 ? g.value : g.unknown_mask;
        r.type = variant_type::ui8;
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<EIGHT_BIT_DISPLAY>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"value"} : n == 1 ? ::fast_io::u8string_view{u8"unknown_mask"} : ::fast_io::u8string_view{};
    }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<EIGHT_BIT_DISPLAY>, EIGHT_BIT_DISPLAY& g,
                                                                                       ::phy_engine::digital::digital_node_update_table&, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        ::std::uint8_t v{}, um{};
        for(int pin = 0; pin < 8; ++pin)
        {
            auto* n = g.pins[pin].nodes;
            s const st = n ? details::read_level(g, n) : s::X;
            auto const bit = static_cast<::std::uint8_t>(1u << static_cast<unsigned>(7 - pin));
            if(st == s::true_state) v |= bit;
            else if(st != s::false_state)
                um |= bit;
        }
        g.value = v;
        g.unknown_mask = um;
        return {};
    }

    // Schmitt trigger (logical/schmitt_trigger.h): an analog input moves the output only across Vth_high (up) / Vth_low (down);
    // a digital input passes through (X / Z -> X); attributes inverted, Vth_low, Vth_high, out (read only)
    struct SCHMITT_TRIGGER
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"SCHMITT_TRIGGER"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"SCHMITT_TRIGGER"};
        pin pins[2]{{{u8"i"}}, {{u8"o"}}};
        double Ll{0.0}, Hl{5.0};
        double Vth_low{1.6666666666666666666666}, Vth_high{3.333333333333333333333};
        bool inverted{};
        digital_node_statement_t last_out{digital_node_statement_t::false_state};
        digital_node_statement_t last_driven{digital_node_statement_t::X};
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<SCHMITT_TRIGGER>, SCHMITT_TRIGGER& g) noexcept { return {g.pins, 2}; }
    inline bool set_attribute_define(model_reserve_type_t<SCHMITT_TRIGGER>, SCHMITT_TRIGGER& g, ::std::size_t n, variant vi) noexcept
    {
        if(n == 0 && vi.type == variant_type::boolean)
        {
            g.inverted = vi.boolean;
            return true;
        }
        if((n == 1 || n == 2) && vi.type == variant_type::d)
        {
            (n == 1 ? g.Vth_low : g.Vth_high) = vi.d;
            return true;
        }
        return false;
    }
    inline variant get_attribute_define(model_reserve_type_t<SCHMITT_TRIGGER>, SCHMITT_TRIGGER const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n == 0)
        {
            r.boolean = g.inverted;
            r.type = variant_type::boolean;
        }
        else if(n == 1 || n == 2)
        {
            r.d = n == 1 ? g.Vth_low : g.Vth_high;
            r.type = variant_type::d;
        }
        else if(n == 3)
        {
            r.digital = g.last_out;
            r.type = variant_type::digital;
        }
        return r;
    }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<SCHMITT_TRIGGER>, ::std::size_t n) noexcept
    {
        constexpr ::fast_io::u8string_view names[4] = {u8"inverted", u8"Vth_low", u8"Vth_high", u8"out"};
        return n < 4 ? names[n] : ::fast_io::u8string_view{};
    }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<SCHMITT_TRIGGER>, SCHMITT_TRIGGER& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        auto* ni = g.pins[0].nodes;
        auto* no = g.pins[1].nodes;
        if(!ni || !no) return {};
        if(ni->num_of_analog_node == 0)
        {
            auto const st = ni->node_information.dn.state;
            g.last_out = details::settled(st) ? st : s::indeterminate_state;
        }
        else
        {
            double const v = ni->node_information.an.voltage.real();
            if(g.last_out == s::false_state)
            {
                if(v >= g.Vth_high) g.last_out = s::true_state;
            }
            else if(g.last_out == s::true_state)
            {
                if(v <= g.Vth_low) g.last_out = s::false_state;
            }
            else
            {
                if(v >= g.Vth_high) g.last_out = s::true_state;
                else if(v <= g.Vth_low)
                    g.last_out = s::false_state;
            }
        }
        s driven = g.last_out;
        if(g.inverted && details::settled(driven)) driven = driven == s::false_state ? s::true_state : s::false_state;
        if(no->num_of_analog_node == 0)
        {
            if(no->node_information.dn.state != driven)
            {
                no->node_information.dn.state = driven;
                table.tables.insert(no);
            }
            g.last_driven = driven;
            return {};
        }
        if(g.last_driven == driven) return {};
        g.last_driven = driven;
        return details::level_of(g, no, driven);
    }

    // ---- four-state helpers of synthesised netlists (logical/resolve2.h, case_eq.h, is_unknown.h, tick_delay.h): these keep Z on
    // their inputs (no Z -> X folding)
    namespace details
    {
        template <typename G>
        inline digital_node_statement_t read_raw(G const& g, node_t* n) noexcept
        {
            using s = digital_node_statement_t;
            if(!n) return s::indeterminate_state;
            if(n->num_of_analog_node == 0) return n->node_information.dn.state;
            double const v = n->node_information.an.voltage.real();
            return v >= g.Hl ? s::true_state : (v <= g.Ll ? s::false_state : s::indeterminate_state);
        }
    }  // namespace details
    template <int KIND>  // 0 RESOLVE2 (a b o: Z yields, equal passes, else X), 1 CASE_EQ (ia ib o: a === b, never X), 2 IS_UNKNOWN (i o: X or Z -> H)
    struct four_state_op
    {
        inline static constexpr ::fast_io::u8string_view names[3] = {u8"RESOLVE2", u8"CASE_EQ", u8"IS_UNKNOWN"};
        inline static constexpr int n_pins = KIND == 2 ? 2 : 3;
        inline static constexpr ::fast_io::u8string_view model_name{names[KIND]};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::update_table};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{names[KIND]};
        pin pins[n_pins]{};
        double Ll{0.0}, Hl{5.0};
        digital_node_statement_t last_out{digital_node_statement_t::X};
        constexpr four_state_op() noexcept
        {
            constexpr char8_t const* pn[3][3] = {{u8"a", u8"b", u8"o"}, {u8"ia", u8"ib", u8"o"}, {u8"i", u8"o"}};
            for(int k = 0; k < n_pins; ++k) pins[k].name = ::fast_io::u8string_view{pn[KIND][k]};
        }
    };
    using RESOLVE2 = four_state_op<0>;
    using CASE_EQ = four_state_op<1>;
    using IS_UNKNOWN = four_state_op<2>;
    template <int KIND>
    inline pin_view generate_pin_view_define(model_reserve_type_t<four_state_op<KIND>>, four_state_op<KIND>& g) noexcept
    {
        return {g.pins, static_cast<::std::size_t>(four_state_op<KIND>::n_pins)};
    }
    template <int KIND>
        requires(KIND != 0)
    inline bool set_attribute_define(model_reserve_type_t<four_state_op<KIND>>, four_state_op<KIND>& g, ::std::size_t n, variant vi) noexcept
    {
        if(vi.type != variant_type::d || n > 1) return false;
        (n == 0 ? g.Ll : g.Hl) = vi.d;
        return true;
    }
    template <int KIND>
        requires(KIND != 0)
    inline variant get_attribute_define(model_reserve_type_t<four_state_op<KIND>>, four_state_op<KIND> const& g, ::std::size_t n) noexcept
    {
        variant r{};
        if(n > 1) return r;
        r.d = n == 0 ? g.Ll : g.Hl;
        r.type = variant_type::d;
        return r;
    }
    template <int KIND>
        requires(KIND != 0)
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<four_state_op<KIND>>, ::std::size_t n) noexcept
    {
        return n == 0 ? ::fast_io::u8string_view{u8"Ll"} : n == 1 ? ::fast_io::u8string_view{u8"Hl"} : ::fast_io::u8string_view{};
    }
    template <int KIND>
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<four_state_op<KIND>>, four_state_op<KIND>& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t) noexcept
    {
        using s = digital_node_statement_t;
        constexpr int NP = four_state_op<KIND>::n_pins;
        for(int k = 0; k < NP; ++k)
            if(!g.pins[k].nodes) return {};
        auto* no = g.pins[NP - 1].nodes;
        auto const a = details::read_raw(g, g.pins[0].nodes);
        s out{};
        if constexpr(KIND == 0)
        {
            auto const b = details::read_raw(g, g.pins[1].nodes);
            out = a == s::high_impedence_state ? b : (b == s::high_impedence_state ? a : (a == b ? a : s::indeterminate_state));  // resolve2.h:50-57
        }
        else if constexpr(KIND == 1)
            out = a == details::read_raw(g, g.pins[1].nodes) ? s::true_state : s::false_state;  // case_eq.h:88
        else
            out = (a == s::indeterminate_state || a == s::high_impedence_state) ? s::true_state : s::false_state;  // is_unknown.h:96-98
        if(no->num_of_analog_node == 0)
        {
            if(no->node_information.dn.state != out)
            {
                no->node_information.dn.state = out;
                table.tables.insert(no);
            }
            g.last_out = out;
            return {};
        }
        if constexpr(KIND == 0) return details::level_of(g, no, out);  // resolve2.h:70-77: drives on every call
        else
        {
            if(g.last_out == out) return {};
            g.last_out = out;
            return details::level_of(g, no, out);
        }
    }

    // tick delay (logical/tick_delay.h): the output is the input of `ticks` digital_clk() calls ago; advances once per tick (in the
    // before-all phase only); the line starts filled with the first input seen; ticks = 0 passes through
    struct TICK_DELAY
    {
        inline static constexpr ::fast_io::u8string_view model_name{u8"TICK_DELAY"};
        inline static constexpr digital_update_method_t digital_update_method{digital_update_method_t::before_all_clk};
        inline static constexpr model_device_type device_type{model_device_type::digital};
        inline static constexpr ::fast_io::u8string_view identification_name{u8"TICK_DELAY"};
        pin pins[2]{{{u8"i"}}, {{u8"o"}}};
        double Ll{0.0}, Hl{5.0};
        ::std::size_t ticks{1};
        ::std::vector<digital_node_statement_t> pipe{};
        digital_node_statement_t last_out{digital_node_statement_t::X};
        TICK_DELAY() = default;
        explicit TICK_DELAY(::std::size_t t) : ticks{t} {}
    };
    inline pin_view generate_pin_view_define(model_reserve_type_t<TICK_DELAY>, TICK_DELAY& g) noexcept { return {g.pins, 2}; }
    inline bool set_attribute_define(model_reserve_type_t<TICK_DELAY>, TICK_DELAY&, ::std::size_t, variant) noexcept { return false; }
    inline variant get_attribute_define(model_reserve_type_t<TICK_DELAY>, TICK_DELAY const&, ::std::size_t) noexcept { return {}; }
    inline ::fast_io::u8string_view get_attribute_name_define(model_reserve_type_t<TICK_DELAY>, ::std::size_t) noexcept { return {}; }
    inline ::phy_engine::digital::need_operate_analog_node_t update_digital_clk_define(model_reserve_type_t<TICK_DELAY>, TICK_DELAY& g,
                                                                                       ::phy_engine::digital::digital_node_update_table& table, double,
                                                                                       digital_update_method_t method) noexcept
    {
        using s = digital_node_statement_t;
        if(method != digital_update_method_t::before_all_clk) return {};
        auto* no = g.pins[1].nodes;
        if(!no) return {};
        auto const in = details::read_raw(g, g.pins[0].nodes);
        s out = in;
        if(g.ticks != 0)
        {
            if(g.pipe.size() != g.ticks)
            {
                g.pipe.assign(g.ticks, in);
                g.last_out = s::indeterminate_state;
            }
            out = g.pipe[g.ticks - 1];
            for(::std::size_t i = g.ticks - 1; i > 0; --i) g.pipe[i] = g.pipe[i - 1];
            g.pipe[0] = in;
        }
        if(no->num_of_analog_node == 0)
        {
            no->node_information.dn.state = out;
            if(g.last_out != out)
            {
                g.last_out = out;
                table.tables.insert(no);
            }
            return {};
        }
        if(out == s::false_state) return {g.Ll, no};
        if(out == s::true_state) return {g.Hl, no};
        return {};
    }

    static_assert(defines::is_valid_digital_model<comparator> && defines::is_valid_digital_model<NOT> && defines::is_valid_digital_model<AND> &&
                  defines::is_valid_digital_model<OUTPUT> && defines::is_valid_digital_model<INPUT> && defines::is_valid_digital_model<TRI> &&
                  defines::is_valid_digital_model<FULL_ADDER> && defines::is_valid_digital_model<MUL2> && defines::is_valid_digital_model<JKFF> &&
                  defines::is_valid_digital_model<COUNTER4> && defines::is_valid_digital_model<DLATCH> && defines::is_valid_digital_model<DFF_ARSTN> &&
                  defines::is_valid_digital_model<RANDOM_GENERATOR4> && defines::is_valid_digital_model<EIGHT_BIT_INPUT> &&
                  defines::is_valid_digital_model<EIGHT_BIT_DISPLAY> && defines::is_valid_digital_model<SCHMITT_TRIGGER> &&
                  defines::is_valid_digital_model<RESOLVE2> && defines::is_valid_digital_model<CASE_EQ> && defines::is_valid_digital_model<IS_UNKNOWN> &&
                  defines::is_valid_digital_model<TICK_DELAY>);
}  // namespace phy_engine::model
