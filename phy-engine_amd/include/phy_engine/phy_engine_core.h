// phy_engine_core.h -- C++ host layer of the MI355X transient engine, API-compatible with the part of Phy-Engine that
// sits around the hot path.  Written from scratch (std containers, no Eigen / absl / fast_io); same namespaces, type
// names, member names and free-function names as the reference so that netlist-building code and tests read the
// same:
//
//   reference                                                        here
//   include/phy_engine/model/{node,pin,branch}/*.h                    model::node_t / pin / branch (+ views)
//   include/phy_engine/model/model_refs/{type,variant,concept,base}.h model concept, ADL *_define hooks, model_base
//   include/phy_engine/netlist/{netlist,operation}.h                  netlist::netlist, add_model/create_node/add_to_node/...
//   include/phy_engine/circuits/{analyze,environment,MNA/mna}.h       analyze_type, environment, MNA::MNA
//   include/phy_engine/circuits/circuit.h:60-1528                     struct circult (sic): analyze/prepare/solve/reset
//
// What differs by design: `circult::analyze()` never stamps on the host.  A model takes part in the GPU-resident
// path through ONE additive hook next to its iterate_*_define hooks,
//       bool gpu_table_define(model_reserve_type_t<M>, M const&, gpu_table_rows&)
// which describes it as rows of the device tables of include/pe_hip.h.  A netlist containing a model without that hook
// is rejected by analyze() (returns false, message in circult::last_error) -- there is no CPU numeric path.
#pragma once
#include <chrono>
#include <cmath>
#include <complex>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <string_view>
#include <type_traits>
#include <vector>

#include <fast_io/fast_io.h>
#include <fast_io/fast_io_dsal/string_view.h>

#include "../../../include/pe_hip.h"

namespace phy_engine
{
    // ---- circuits/analyze.h:7-16
    enum class analyze_type : ::std::uint_fast32_t
    {
        OP = 0,
        DC,
        AC,
        ACOP,
        TR,
        TROP
    };

    // ---- circuits/environment/environment.h:7-22
    struct environment
    {
        double V_eps_max{};
        double V_epsr_max{};
        double I_eps_max{};
        double I_epsr_max{};
        double charge_eps_max{};
        double g_min{};
        double r_open{};
        double t_TOEF{};
        double temperature{27.0};
        double norm_temperature{27.0};
    };

    namespace analyzer
    {
        struct TR
        {
            double t_stop{};
            double t_step{};
        };
        struct DC  // circuits/analyzer/DC.h:5-8
        {
            double m_currentOmega{};
        };
        struct AC  // circuits/analyzer/AC.h
        {
            enum class sweep_type : ::std::uint_fast8_t
            {
                single = 0,
                linear = 1,
                log = 2,
            };
            sweep_type sweep{sweep_type::single};
            double omega{};        // rad/s; updated per point in sweep mode
            double omega_start{};
            double omega_stop{};
            ::std::size_t points{};
        };
        struct analyzer_storage_t
        {
            TR tr{};
            DC dc{};
            AC ac{};
        };
    }  // namespace analyzer

    namespace model
    {
        struct node_t;
        struct model_base;

        // ---- model/model_refs/type.h
        enum class model_type : ::std::size_t
        {
            null,
            invalid,
            normal
        };
        enum class model_device_type : ::std::uint_fast8_t
        {
            linear,
            non_linear,
            digital
        };

        // ---- model/node/node.h:25-36 (4-state logic) and :255-306
        enum class digital_node_statement_t : ::std::uint_fast8_t
        {
            false_state = 0,
            true_state = 1,
            indeterminate_state = 2,
            high_impedence_state = 3,
            L = false_state,
            H = true_state,
            X = indeterminate_state,
            Z = high_impedence_state
        };

        struct pin
        {
            ::fast_io::u8string_view name{};
            node_t* nodes{};
            model_base* model{};
        };
        struct pin_view
        {
            pin* pins{};
            ::std::size_t size{};
        };
        struct branch
        {
            ::std::size_t index{};
            ::std::complex<double> current{};
        };
        struct branch_view
        {
            branch* branches{};
            ::std::size_t size{};
        };

        struct analog_node_t
        {
            ::std::complex<double> voltage{};
        };
        struct digital_node_t
        {
            digital_node_statement_t state{};
        };
        union node_information_union
        {
            analog_node_t an;
            digital_node_t dn;
            node_information_union() : an{} {}
        };

        struct node_t
        {
            node_information_union node_information{};
            ::std::set<pin*> pins{};
            ::std::size_t num_of_analog_node{};
            ::std::size_t node_index{SIZE_MAX};

            node_t() = default;
            node_t(node_t const& o) : node_information{o.node_information} {}
            node_t& operator=(node_t const& o)
            {
                node_information = o.node_information;
                pins.clear();
                return *this;
            }
            ~node_t() { clear(); }
            void destroy() noexcept
            {
                pins.clear();
                num_of_analog_node = 0;
                node_index = SIZE_MAX;
            }
            void clear() noexcept
            {
                for(auto* p: pins) p->nodes = nullptr;
                destroy();
            }
        };
        struct node_view
        {
            node_t* nodes{};
            ::std::size_t size{};
        };

        // ---- model/node/node.h:18-23
        enum class digital_update_method_t : ::std::uint_fast8_t
        {
            update_table = 0x00,
            before_all_clk = 0x01,
            after_all_clk = 0x02,
        };

        // 4-state algebra (model/node/node.h:78-230): L dominates AND, H dominates OR, X / Z otherwise unknown
        inline constexpr digital_node_statement_t operator&(digital_node_statement_t a, digital_node_statement_t b) noexcept
        {
            using s = digital_node_statement_t;
            if(a == s::false_state || b == s::false_state) return s::false_state;
            if(a == s::true_state && b == s::true_state) return s::true_state;
            return s::indeterminate_state;
        }
        inline constexpr digital_node_statement_t operator|(digital_node_statement_t a, digital_node_statement_t b) noexcept
        {
            using s = digital_node_statement_t;
            if(a == s::true_state || b == s::true_state) return s::true_state;
            if(a == s::false_state && b == s::false_state) return s::false_state;
            return s::indeterminate_state;
        }
        inline constexpr digital_node_statement_t operator~(digital_node_statement_t a) noexcept
        {
            using s = digital_node_statement_t;
            return a == s::false_state ? s::true_state : (a == s::true_state ? s::false_state : s::indeterminate_state);
        }
        inline constexpr digital_node_statement_t operator^(digital_node_statement_t a, digital_node_statement_t b) noexcept
        {
            using s = digital_node_statement_t;
            bool const da = a == s::false_state || a == s::true_state, db = b == s::false_state || b == s::true_state;
            if(!da || !db) return s::indeterminate_state;
            return (a == s::true_state) != (b == s::true_state) ? s::true_state : s::false_state;
        }

        // ---- model/model_refs/variant.h
        enum class variant_type : ::std::uint_fast8_t
        {
            invalid,
            i8,
            i16,
            i32,
            i64,
            ui8,
            ui16,
            ui32,
            ui64,
            boolean,
            f,
            d,
            digital
        };
        struct variant
        {
            union
            {
                ::std::int_least8_t i8;
                ::std::int_least16_t i16;
                ::std::int_least32_t i32;
                ::std::int_least64_t i64;
                ::std::uint_least8_t ui8;
                ::std::uint_least16_t ui16;
                ::std::uint_least32_t ui32;
                ::std::uint_least64_t ui64{};
                bool boolean;
                float f;
                double d;
                digital_node_statement_t digital;
            };
            variant_type type{};
        };
    }  // namespace model

    // ---- circuits/MNA/mna.h:12-169: the sparse system the stamp hooks write into -- rows of (column -> value) maps plus the
    // right-hand side Z, with the G / B / C / D / I / E views over [nodes | branches] and the "index SIZE_MAX = ground, write
    // into a scratch cell" rule (mna.h:62).  On the MI355X path the built-in models never touch it (their stamps are device
    // tables); it carries (1) the stamps of plug-in models that only have host hooks (the overlay circult::prepare builds and
    // the engine adds every Newton iteration), and (2) after an analysis of a small circuit, a copy of the last assembled
    // system for inspection (circult::mna, as tests of the reference dump it).
    namespace MNA
    {
        struct MNA
        {
            using value_type = ::std::complex<double>;
            using row_type = ::std::map<::std::size_t, value_type>;

            MNA() noexcept = default;
            MNA(::std::size_t ns, ::std::size_t bs) : node_size{ns}, branch_size{bs} { A.resize(ns + bs); }
            void resize(::std::size_t ns, ::std::size_t bs)
            {
                node_size = ns;
                branch_size = bs;
                A.resize(ns + bs);
            }
            void clear() noexcept
            {
                for(auto& row: A) row.clear();
                Z.clear();
            }
            // keeps the cells (the pattern only ever grows), zeroes the values
            void clear_values_keep_pattern() noexcept
            {
                for(auto& row: A)
                    for(auto& kv: row) kv.second = {};
                for(auto& kv: Z) kv.second = {};
            }
            void clear_destroy() noexcept { clear(); }

            value_type& A_ref(::std::size_t row, ::std::size_t col) { return (row == SIZE_MAX || col == SIZE_MAX) ? scratch_ : A[row][col]; }
            value_type& G_ref(::std::size_t row, ::std::size_t col) { return A_ref(row, col); }
            value_type& B_ref(::std::size_t row, ::std::size_t col) { return (row == SIZE_MAX || col == SIZE_MAX) ? scratch_ : A[row][col + node_size]; }
            value_type& C_ref(::std::size_t row, ::std::size_t col) { return (row == SIZE_MAX || col == SIZE_MAX) ? scratch_ : A[row + node_size][col]; }
            value_type& D_ref(::std::size_t row, ::std::size_t col)
            {
                return (row == SIZE_MAX || col == SIZE_MAX) ? scratch_ : A[row + node_size][col + node_size];
            }
            value_type& Z_ref(::std::size_t row) { return row == SIZE_MAX ? scratch_ : Z[row]; }
            value_type& I_ref(::std::size_t row) { return Z_ref(row); }
            value_type& E_ref(::std::size_t row) { return row == SIZE_MAX ? scratch_ : Z[row + node_size]; }

            ::std::vector<row_type> A{};
            row_type Z{};
            ::std::size_t node_size{};
            ::std::size_t branch_size{};
            double r_open{1e12};

        private:
            value_type scratch_{};
        };
    }  // namespace MNA

    // ---- circuits/digital/update_table.h:8-27
    namespace digital
    {
        struct digital_node_update_table
        {
            ::std::set<::phy_engine::model::node_t*> always_tables{};  // hybrid nodes: re-evaluated on every tick
            ::std::set<::phy_engine::model::node_t*> tables{};         // pending nodes of the current tick
        };
        struct need_operate_analog_node_t
        {
            double voltage{};
            ::phy_engine::model::node_t* need_to_operate_analog_node{};
        };
    }  // namespace digital

    namespace model
    {
        // ---- the additive device-table hook ---------------------------------------------------------------------
        struct gpu_table_row
        {
            int kind{};               // pe_hip_kind
            int pin_a{}, pin_b{};     // indices into the model's pin view
            int branch{-1};           // index into the model's branch view (kinds with a branch row), else -1
            double params[PE_HIP_DIODE_NPARAM]{};
            int pin_c{-1}, pin_d{-1}; // third / fourth pin of the four-pin kinds (pe_hip.h)
            int branch2{-1};          // second branch row of the two-branch kinds
            int pin_e{-1};            // fifth pin / third branch row (center-tap transformer)
            int branch3{-1};
        };
        // node / branch / parameter columns per device of a pe_hip_kind (pe_hip.h)
        inline constexpr int gpu_kind_pins(int k) noexcept
        {
            if(k == PE_HIP_XFMR_CT) return 5;
            if(k == PE_HIP_RELAY) return 4;
            return k >= PE_HIP_NMOS ? 3 : ((k <= PE_HIP_IAC || k == PE_HIP_SWITCH || k == PE_HIP_VGEN) ? 2 : 4);
        }
        inline constexpr int gpu_kind_branches(int k) noexcept
        {
            switch(k)
            {
                case PE_HIP_L:
                case PE_HIP_VDC:
                case PE_HIP_VAC:
                case PE_HIP_VCVS:
                case PE_HIP_CCCS:
                case PE_HIP_OPAMP:
                case PE_HIP_SWITCH:
                case PE_HIP_RELAY:
                case PE_HIP_VGEN: return 1;
                case PE_HIP_XFMR_CT: return 3;
                case PE_HIP_CCVS:
                case PE_HIP_XFMR:
                case PE_HIP_COUPLED_L: return 2;
                default: return 0;
            }
        }
        inline constexpr int gpu_kind_ncol(int k) noexcept
        {
            switch(k)
            {
                case PE_HIP_VAC:
                case PE_HIP_IAC:
                case PE_HIP_COUPLED_L:
                case PE_HIP_NMOS:
                case PE_HIP_PMOS: return 3;
                case PE_HIP_BJT_NPN:
                case PE_HIP_BJT_PNP: return 5;
                case PE_HIP_RELAY: return 2;
                case PE_HIP_DIODE: return PE_HIP_DIODE_NPARAM;
                case PE_HIP_VGEN: return PE_HIP_VGEN_NPARAM;
                default: return 1;
            }
        }
        struct gpu_table_rows
        {
            gpu_table_row row[4]{};
            int count{};
        };

        // ---- model/model_refs/concept.h:23-217 (same hook names and signatures)
        template <typename mod>
        struct model_reserve_type_t
        {
            static_assert(::std::is_same_v<::std::remove_cvref_t<mod>, mod>);
            explicit constexpr model_reserve_type_t() noexcept = default;
        };
        template <typename mod>
        inline constexpr model_reserve_type_t<mod> model_reserve_type{};

        namespace defines
        {
            template <typename mod>
            concept can_prepare_foundation = requires(mod&& t) {
                { prepare_foundation_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_init = requires(mod&& t) {
                { init_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_prepare_highest_priority = requires(mod&& t) {
                { prepare_highest_priority_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_prepare_ac = requires(mod&& t) {
                { prepare_ac_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_prepare_dc = requires(mod&& t) {
                { prepare_dc_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_prepare_tr = requires(mod&& t) {
                { prepare_tr_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_prepare_op = requires(mod&& t) {
                { prepare_op_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_prepare_trop = requires(mod&& t) {
                { prepare_trop_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_save_op = requires(mod&& t) {
                { save_op_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_load_temperature = requires(mod&& t) {
                { load_temperature_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, double{}) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_adapt_step = requires(mod&& t, double step) {
                { adapt_step_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, step) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_check_convergence = requires(mod&& t) {
                { check_convergence_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_query_status = requires(mod&& t) {
                { query_status_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, ::std::size_t{}) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_iterate_ac = requires(mod&& t, ::phy_engine::MNA::MNA& mna) {
                { iterate_ac_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, mna, double{}) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_iterate_dc = requires(mod&& t, ::phy_engine::MNA::MNA& mna) {
                { iterate_dc_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, mna) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_iterate_tr = requires(mod&& t, ::phy_engine::MNA::MNA& mna) {
                { iterate_tr_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, mna, double{}) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_iterate_op = requires(mod&& t, ::phy_engine::MNA::MNA& mna) {
                { iterate_op_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, mna) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_iterate_trop = requires(mod&& t, ::phy_engine::MNA::MNA& mna) {
                { iterate_trop_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, mna) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept can_step_changed_tr = requires(mod&& t) {
                { step_changed_tr_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, double{}, double{}) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept has_set_attribute = requires(mod&& t) {
                { set_attribute_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, ::std::size_t{}, variant{}) } -> ::std::same_as<bool>;
            };
            template <typename mod>
            concept has_get_attribute = requires(mod&& t) {
                { get_attribute_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, ::std::size_t{}) } -> ::std::same_as<variant>;
            };
            template <typename mod>
            concept has_full_get_attribute_name = requires(mod&& t) {
                { get_attribute_name_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, ::std::size_t{}) } -> ::std::same_as<::fast_io::u8string_view>;
            };
            template <typename mod>
            concept has_reduced_get_attribute_name = requires() {
                { get_attribute_name_define(model_reserve_type<::std::remove_cvref_t<mod>>, ::std::size_t{}) } -> ::std::same_as<::fast_io::u8string_view>;
            };
            template <typename mod>
            concept has_get_attribute_name = has_full_get_attribute_name<mod> || has_reduced_get_attribute_name<mod>;
            template <typename mod>
            concept has_attribute = has_set_attribute<mod> && has_get_attribute<mod> && has_get_attribute_name<mod>;
            template <typename mod>
            concept can_generate_pin_view = requires(mod&& t) {
                { generate_pin_view_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<pin_view>;
            };
            template <typename mod>
            concept can_generate_branch_view = requires(mod&& t) {
                { generate_branch_view_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<branch_view>;
            };
            template <typename mod>
            concept can_generate_internal_node_view = requires(mod&& t) {
                { generate_internal_node_define(model_reserve_type<::std::remove_cvref_t<mod>>, t) } -> ::std::same_as<node_view>;
            };
            template <typename mod>
            concept can_gpu_table = requires(mod const& t, gpu_table_rows& rows) {
                { gpu_table_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, rows) } -> ::std::same_as<bool>;
            };
            // a model the HOST can stamp: the reference's own notion of can_iterate_mna (concept.h:183)
            template <typename mod>
            concept can_host_stamp = can_iterate_ac<mod> || can_iterate_dc<mod> || can_iterate_op<mod> || can_iterate_tr<mod> || can_iterate_trop<mod>;
            template <typename mod>
            concept can_iterate_mna = can_host_stamp<mod> || can_gpu_table<mod>;
            template <typename mod>
            concept can_update_digital_clk = requires(mod&& t, ::phy_engine::digital::digital_node_update_table& table, double tr_duration, digital_update_method_t method) {
                {
                    update_digital_clk_define(model_reserve_type<::std::remove_cvref_t<mod>>, t, table, tr_duration, method)
                } -> ::std::same_as<::phy_engine::digital::need_operate_analog_node_t>;
            };
            template <typename mod>
            concept has_digital_update_method = ::std::same_as<::std::remove_cvref_t<decltype(::std::remove_cvref_t<mod>::digital_update_method)>, digital_update_method_t>;
            template <typename mod>
            concept is_valid_digital_model = ::std::remove_cvref_t<mod>::device_type == model_device_type::digital && can_update_digital_clk<mod> && has_digital_update_method<mod>;
        }  // namespace defines

        namespace details
        {
            template <typename mod, typename = void>
            struct model_check : ::std::false_type
            {
            };
            template <typename mod>
            struct model_check<mod,
                               ::std::void_t<decltype(::std::remove_cvref_t<mod>::model_name), decltype(::std::remove_cvref_t<mod>::device_type),
                                             decltype(::std::remove_cvref_t<mod>::identification_name)>> :
                ::std::bool_constant<::std::is_same_v<::std::remove_cvref_t<decltype(::std::remove_cvref_t<mod>::model_name)>, ::fast_io::u8string_view> &&
                                     ::std::is_same_v<::std::remove_cvref_t<decltype(::std::remove_cvref_t<mod>::device_type)>, model_device_type> &&
                                     ::std::is_same_v<::std::remove_cvref_t<decltype(::std::remove_cvref_t<mod>::identification_name)>, ::fast_io::u8string_view>>
            {
            };

            // type-erased interface (model/model_refs/base.h:21-62), reduced to what the resident path consults
            struct model_base_impl
            {
                virtual ~model_base_impl() = default;
                virtual model_base_impl* clone() const = 0;
                virtual bool set_attribute(::std::size_t index, variant vi) noexcept = 0;
                virtual variant get_attribute(::std::size_t index) noexcept = 0;
                virtual ::fast_io::u8string_view get_attribute_name(::std::size_t index) noexcept = 0;
                virtual pin_view generate_pin_view() noexcept = 0;
                virtual branch_view generate_branch_view() noexcept = 0;
                virtual node_view generate_internal_node_view() noexcept = 0;
                virtual ::fast_io::u8string_view get_model_name() noexcept = 0;
                virtual ::fast_io::u8string_view get_identification_name() noexcept = 0;
                virtual model_device_type get_device_type() noexcept = 0;
                virtual bool has_gpu_table() noexcept = 0;
                virtual bool gpu_table(gpu_table_rows& rows) noexcept = 0;
                // host hooks with the fallback chains of model/model_refs/base.h:103-330 (used for models without a device table)
                virtual bool has_host_stamp() noexcept = 0;
                virtual bool init_model() noexcept = 0;
                virtual bool prepare_for(int analysis) noexcept = 0;  // analyze_type value: prepare_{op,dc,ac,tr,trop} + fallbacks
                virtual bool iterate_ac(::phy_engine::MNA::MNA& mna, double omega) noexcept = 0;
                virtual bool iterate_dc(::phy_engine::MNA::MNA& mna) noexcept = 0;
                virtual bool iterate_tr(::phy_engine::MNA::MNA& mna, double t_time) noexcept = 0;
                virtual bool iterate_op(::phy_engine::MNA::MNA& mna) noexcept = 0;
                virtual bool iterate_trop(::phy_engine::MNA::MNA& mna) noexcept = 0;
                virtual bool save_op() noexcept = 0;
                virtual bool load_temperature(double temp) noexcept = 0;
                virtual ::std::size_t find_attribute(::std::u8string_view lower_case_name) noexcept = 0;
                virtual bool step_changed_tr(double last_step, double now_step) noexcept = 0;
                virtual bool adapt_step(double& step) noexcept = 0;
                virtual bool check_convergence() noexcept = 0;
                virtual ::phy_engine::digital::need_operate_analog_node_t
                    update_digital_clk(::phy_engine::digital::digital_node_update_table& table, double tr_duration, digital_update_method_t method) noexcept = 0;
                virtual digital_update_method_t get_digital_update_method() noexcept = 0;
            };

            template <typename mod>
            struct model_derv_impl final : model_base_impl
            {
                using T = ::std::remove_cvref_t<mod>;
                T m;
                explicit model_derv_impl(T const& in) : m{in} {}
                explicit model_derv_impl(T&& in) : m{::std::move(in)} {}
                model_base_impl* clone() const override { return new model_derv_impl<T>{m}; }
                bool set_attribute(::std::size_t index, variant vi) noexcept override
                {
                    if constexpr(defines::has_set_attribute<T>) return set_attribute_define(model_reserve_type<T>, m, index, vi);
                    else
                        return false;
                }
                variant get_attribute(::std::size_t index) noexcept override
                {
                    if constexpr(defines::has_get_attribute<T>) return get_attribute_define(model_reserve_type<T>, m, index);
                    else
                        return {};
                }
                ::fast_io::u8string_view get_attribute_name(::std::size_t index) noexcept override
                {
                    if constexpr(defines::has_full_get_attribute_name<T>) return get_attribute_name_define(model_reserve_type<T>, m, index);
                    else if constexpr(defines::has_reduced_get_attribute_name<T>)
                        return get_attribute_name_define(model_reserve_type<T>, index);
                    else
                        return {};
                }
                pin_view generate_pin_view() noexcept override { return generate_pin_view_define(model_reserve_type<T>, m); }
                branch_view generate_branch_view() noexcept override
                {
                    if constexpr(defines::can_generate_branch_view<T>) return generate_branch_view_define(model_reserve_type<T>, m);
                    else
                        return {};
                }
                node_view generate_internal_node_view() noexcept override
                {
                    if constexpr(defines::can_generate_internal_node_view<T>) return generate_internal_node_define(model_reserve_type<T>, m);
                    else
                        return {};
                }
                ::fast_io::u8string_view get_model_name() noexcept override { return T::model_name; }
                ::fast_io::u8string_view get_identification_name() noexcept override { return T::identification_name; }
                model_device_type get_device_type() noexcept override { return T::device_type; }
                bool has_gpu_table() noexcept override { return defines::can_gpu_table<T>; }
                bool gpu_table(gpu_table_rows& rows) noexcept override
                {
                    if constexpr(defines::can_gpu_table<T>) return gpu_table_define(model_reserve_type<T>, m, rows);
                    else
                        return false;
                }
                bool has_host_stamp() noexcept override { return defines::can_host_stamp<T>; }
                bool init_model() noexcept override
                {
                    if constexpr(defines::can_init<T>) return init_define(model_reserve_type<T>, m);
                    else
                        return true;
                }
                // base.h:103-215: highest priority first, then the analysis' own hook, its documented fallback, the foundation
                bool prepare_for(int analysis) noexcept override
                {
                    constexpr auto tag = model_reserve_type<T>;
                    if constexpr(defines::can_prepare_highest_priority<T>) return prepare_highest_priority_define(tag, m);
                    else
                    {
                        auto foundation = [&]() -> bool
                        {
                            if constexpr(defines::can_prepare_foundation<T>) return prepare_foundation_define(tag, m);
                            else
                                return true;
                        };
                        switch(static_cast<analyze_type>(analysis))
                        {
                            case analyze_type::AC:
                            case analyze_type::ACOP:
                                if constexpr(defines::can_prepare_ac<T>) return prepare_ac_define(tag, m);
                                else
                                    return foundation();
                            case analyze_type::TR:
                                if constexpr(defines::can_prepare_tr<T>) return prepare_tr_define(tag, m);
                                else
                                    return foundation();
                            case analyze_type::OP:
                                if constexpr(defines::can_prepare_op<T>) return prepare_op_define(tag, m);
                                else if constexpr(defines::can_prepare_dc<T>)
                                    return prepare_dc_define(tag, m);
                                else
                                    return foundation();
                            case analyze_type::TROP:
                                if constexpr(defines::can_prepare_trop<T>) return prepare_trop_define(tag, m);
                                else if constexpr(defines::can_prepare_tr<T>)
                                    return prepare_tr_define(tag, m);
                                else
                                    return foundation();
                            default:
                                if constexpr(defines::can_prepare_dc<T>) return prepare_dc_define(tag, m);
                                else
                                    return foundation();
                        }
                    }
                }
                // base.h:216-304: ac -> dc; tr -> dc; op -> dc; trop -> tr(0) -> dc; a model with some other stamp hook stamps nothing
                bool iterate_dc(::phy_engine::MNA::MNA& mna) noexcept override
                {
                    if constexpr(defines::can_iterate_dc<T>) return iterate_dc_define(model_reserve_type<T>, m, mna);
                    else
                        return defines::can_iterate_mna<T> || defines::is_valid_digital_model<T>;
                }
                bool iterate_ac(::phy_engine::MNA::MNA& mna, double omega) noexcept override
                {
                    if constexpr(defines::can_iterate_ac<T>) return iterate_ac_define(model_reserve_type<T>, m, mna, omega);
                    else
                        return iterate_dc(mna);
                }
                bool iterate_tr(::phy_engine::MNA::MNA& mna, double t_time) noexcept override
                {
                    if constexpr(defines::can_iterate_tr<T>) return iterate_tr_define(model_reserve_type<T>, m, mna, t_time);
                    else
                        return iterate_dc(mna);
                }
                bool iterate_op(::phy_engine::MNA::MNA& mna) noexcept override
                {
                    if constexpr(defines::can_iterate_op<T>) return iterate_op_define(model_reserve_type<T>, m, mna);
                    else
                        return iterate_dc(mna);
                }
                bool iterate_trop(::phy_engine::MNA::MNA& mna) noexcept override
                {
                    if constexpr(defines::can_iterate_trop<T>) return iterate_trop_define(model_reserve_type<T>, m, mna);
                    else if constexpr(defines::can_iterate_tr<T>)
                        return iterate_tr_define(model_reserve_type<T>, m, mna, 0.0);
                    else
                        return iterate_dc(mna);
                }
                bool save_op() noexcept override
                {
                    if constexpr(T::device_type == model_device_type::non_linear && defines::can_save_op<T>) return save_op_define(model_reserve_type<T>, m);
                    else
                        return true;
                }
                // base.h:326-380: without a hook of its own, a model that exposes a numeric attribute called "Temp" (any case) takes
                // the environment temperature through set_attribute and re-derives its foundation quantities
                bool load_temperature(double temp) noexcept override
                {
                    if constexpr(defines::can_load_temperature<T>) return load_temperature_define(model_reserve_type<T>, m, temp);
                    else
                    {
                        if constexpr(defines::has_set_attribute<T> && defines::has_get_attribute_name<T>)
                        {
                            ::std::size_t const idx{find_attribute(u8"temp")};
                            if(idx != SIZE_MAX)
                            {
                                variant v{};
                                v.d = temp;
                                v.type = variant_type::d;
                                (void)set_attribute(idx, v);
                                if constexpr(defines::can_prepare_foundation<T>) (void)prepare_foundation_define(model_reserve_type<T>, m);
                            }
                        }
                        return true;
                    }
                }
                // first attribute whose name equals `lower` ignoring ASCII case (the scan limits of base.h:354-372), or SIZE_MAX
                ::std::size_t find_attribute(::std::u8string_view lower) noexcept override
                {
                    ::std::size_t empty_run{};
                    bool seen{};
                    for(::std::size_t idx{}; idx < 512; ++idx)
                    {
                        auto const n = get_attribute_name(idx);
                        if(n.empty())
                        {
                            if(seen && ++empty_run >= 64) break;
                            continue;
                        }
                        seen = true;
                        empty_run = 0;
                        if(n.size() != lower.size()) continue;
                        bool same{true};
                        for(::std::size_t i{}; i < lower.size() && same; ++i)
                        {
                            auto ch = static_cast<unsigned char>(n[i]);
                            if(ch >= 'A' && ch <= 'Z') ch = static_cast<unsigned char>(ch + ('a' - 'A'));
                            same = ch == static_cast<unsigned char>(lower[i]);
                        }
                        if(same) return idx;
                    }
                    return SIZE_MAX;
                }
                bool step_changed_tr(double last_step, double now_step) noexcept override
                {
                    if constexpr(defines::can_step_changed_tr<T>) return step_changed_tr_define(model_reserve_type<T>, m, last_step, now_step);
                    else
                        return true;
                }
                bool adapt_step(double& step) noexcept override
                {
                    if constexpr(defines::can_adapt_step<T>) return adapt_step_define(model_reserve_type<T>, m, step);
                    else
                        return true;
                }
                bool check_convergence() noexcept override
                {
                    if constexpr(defines::can_check_convergence<T>) return check_convergence_define(model_reserve_type<T>, m);
                    else
                        return true;
                }
                ::phy_engine::digital::need_operate_analog_node_t
                    update_digital_clk(::phy_engine::digital::digital_node_update_table& table, double tr_duration, digital_update_method_t method) noexcept override
                {
                    if constexpr(defines::can_update_digital_clk<T>) return update_digital_clk_define(model_reserve_type<T>, m, table, tr_duration, method);
                    else
                        return {};
                }
                digital_update_method_t get_digital_update_method() noexcept override
                {
                    if constexpr(defines::has_digital_update_method<T>) return T::digital_update_method;
                    else
                        return digital_update_method_t::update_table;
                }
            };
        }  // namespace details

        template <typename mod>
        concept model = details::model_check<mod>::value;

        // ---- owning handle (model/model_refs/base.h:534-827)
        struct model_base
        {
            model_type type{};
            details::model_base_impl* ptr{};
            ::std::size_t identification{};
            ::std::u8string name{};
            ::std::u8string describe{};
            bool has_init{};

            model_base() = default;
            template <typename mod>
                requires (model<mod> && !::std::is_same_v<::std::remove_cvref_t<mod>, model_base>)
            model_base(mod&& m) : type{model_type::normal}, ptr{new details::model_derv_impl<::std::remove_cvref_t<mod>>{::std::forward<mod>(m)}}
            {
            }
            model_base(model_base const& o) : type{o.type}, ptr{o.ptr ? o.ptr->clone() : nullptr}, identification{o.identification}, name{o.name}, describe{o.describe}, has_init{o.has_init}
            {
                detach_pins();
            }
            model_base(model_base&& o) noexcept : type{o.type}, ptr{o.ptr}, identification{o.identification}, name{::std::move(o.name)}, describe{::std::move(o.describe)}, has_init{o.has_init}
            {
                o.ptr = nullptr;
                o.type = model_type::null;
            }
            model_base& operator=(model_base const& o)
            {
                if(this == &o) return *this;
                delete ptr;
                type = o.type;
                ptr = o.ptr ? o.ptr->clone() : nullptr;
                identification = o.identification;
                name = o.name;
                describe = o.describe;
                has_init = o.has_init;
                detach_pins();
                return *this;
            }
            ~model_base() { clear(); }
            void clear() noexcept
            {
                if(ptr)
                {
                    auto pv = ptr->generate_pin_view();
                    for(::std::size_t i = 0; i < pv.size; ++i)
                        if(pv.pins[i].nodes)
                        {
                            pv.pins[i].nodes->pins.erase(pv.pins + i);
                            if(ptr->get_device_type() != model_device_type::digital && pv.pins[i].nodes->num_of_analog_node) --pv.pins[i].nodes->num_of_analog_node;
                            pv.pins[i].nodes = nullptr;
                        }
                    delete ptr;
                    ptr = nullptr;
                }
                type = model_type::null;
            }

        private:
            void detach_pins() noexcept
            {
                if(!ptr) return;
                auto pv = ptr->generate_pin_view();
                for(::std::size_t i = 0; i < pv.size; ++i) pv.pins[i].nodes = nullptr;
            }
        };
    }  // namespace model

    // ---- netlist/netlist.h + operation.h: chunked arenas (addresses of models and nodes are stable), same operations
    namespace netlist
    {
        namespace details
        {
            template <typename T>
            struct block
            {
                inline static constexpr ::std::size_t chunk_size{4096};
                inline static constexpr ::std::size_t chunk_module_size{chunk_size / sizeof(T) > 0 ? chunk_size / sizeof(T) : 1};
                T* begin{};
                T* curr{};
                ::std::size_t num_of_null_model{};
                block() : begin{static_cast<T*>(::operator new(sizeof(T) * chunk_module_size))}, curr{begin} {}
                block(block&& o) noexcept : begin{o.begin}, curr{o.curr}, num_of_null_model{o.num_of_null_model} { o.begin = o.curr = nullptr; }
                block(block const&) = delete;
                ~block()
                {
                    for(T* p = begin; p != curr; ++p) p->~T();
                    ::operator delete(begin);
                }
                ::std::size_t size() const noexcept { return static_cast<::std::size_t>(curr - begin); }
            };
            using netlist_model_base_block = block<::phy_engine::model::model_base>;
            using netlist_node_block = block<::phy_engine::model::node_t>;
        }  // namespace details

        struct netlist
        {
            ::std::vector<details::netlist_model_base_block> models{};
            ::std::vector<details::netlist_node_block> nodes{};
            ::phy_engine::model::node_t ground_node{};
            netlist() = default;
            // netlist/netlist.h:182-329: deep copy.  Models are cloned chunk by chunk at the same (vec_pos, chunk_pos) positions,
            // nodes likewise, and every pin of a copied model is attached to the copy of the node it was attached to.
            netlist(netlist const& o) { copy_from(o); }
            netlist& operator=(netlist const& o)
            {
                if(this != &o)
                {
                    models.clear();
                    nodes.clear();
                    ground_node.clear();
                    copy_from(o);
                }
                return *this;
            }
            netlist(netlist&&) = delete;  // (addresses of models and nodes are what pins and callers hold)

        private:
            void copy_from(netlist const& o)
            {
                ::std::map<::phy_engine::model::node_t const*, ::phy_engine::model::node_t*> twin;
                twin[&o.ground_node] = &ground_node;
                ground_node.node_information = o.ground_node.node_information;
                ground_node.num_of_analog_node = o.ground_node.num_of_analog_node;
                nodes.reserve(o.nodes.size());
                for(auto const& ob: o.nodes)
                {
                    auto& nb = nodes.emplace_back();
                    for(auto const* p = ob.begin; p != ob.curr; ++p, ++nb.curr)
                    {
                        ::new(nb.curr)::phy_engine::model::node_t{*p};  // (node_t's copy keeps the state, not the pin set)
                        nb.curr->num_of_analog_node = p->num_of_analog_node;
                        twin[p] = nb.curr;
                    }
                }
                models.reserve(o.models.size());
                for(auto const& ob: o.models)
                {
                    auto& nb = models.emplace_back();
                    nb.num_of_null_model = ob.num_of_null_model;
                    for(auto const* p = ob.begin; p != ob.curr; ++p, ++nb.curr)
                    {
                        ::new(nb.curr)::phy_engine::model::model_base{*p};  // clone; its pins come back detached
                        if(p->type != ::phy_engine::model::model_type::normal || !p->ptr) continue;
                        auto const from = p->ptr->generate_pin_view();
                        auto const to = nb.curr->ptr->generate_pin_view();
                        for(::std::size_t i = 0; i < from.size && i < to.size; ++i)
                        {
                            auto const it = from.pins[i].nodes ? twin.find(from.pins[i].nodes) : twin.end();
                            if(it == twin.end()) continue;  // unattached, or attached to a node of another netlist
                            to.pins[i].nodes = it->second;
                            it->second->pins.insert(to.pins + i);
                        }
                    }
                }
            }

        public:
        };

        struct model_pos
        {
            ::std::size_t vec_pos{};
            ::std::size_t chunk_pos{};
        };
        struct add_model_retstr
        {
            ::phy_engine::model::model_base* mod{};
            model_pos mod_pos{};
        };

        inline ::phy_engine::model::node_t& get_ground_node(netlist& nl) noexcept { return nl.ground_node; }

        // netlist/operation.h:46-89
        template <typename mod>
            requires (::phy_engine::model::model<mod> && ::phy_engine::model::defines::can_generate_pin_view<mod> &&
                      (::phy_engine::model::defines::can_iterate_mna<mod> || ::phy_engine::model::defines::is_valid_digital_model<mod>))
        inline add_model_retstr add_model(netlist& nl, mod&& m)
        {
            if(nl.models.empty() || nl.models.back().size() == details::netlist_model_base_block::chunk_module_size) nl.models.emplace_back();
            auto& blk = nl.models.back();
            ::new(blk.curr)::phy_engine::model::model_base{::std::forward<mod>(m)};
            add_model_retstr r{blk.curr, {blk.size(), nl.models.size() - 1}};
            ++blk.curr;
            return r;
        }

        inline ::phy_engine::model::model_base* get_model(netlist const& nl, ::std::size_t vec_pos, ::std::size_t chunk_pos) noexcept
        {
            if(chunk_pos >= nl.models.size()) return nullptr;
            auto const& blk = nl.models[chunk_pos];
            if(vec_pos >= blk.size()) return nullptr;
            return blk.begin + vec_pos;
        }
        inline ::phy_engine::model::model_base* get_model(netlist const& nl, model_pos pos) noexcept { return get_model(nl, pos.vec_pos, pos.chunk_pos); }

        // netlist/operation.h:91-133: the last model of a chunk is destroyed, others become null placeholders
        inline bool delete_model(netlist& nl, ::std::size_t vec_pos, ::std::size_t chunk_pos) noexcept
        {
            if(chunk_pos >= nl.models.size()) return false;
            auto& blk = nl.models[chunk_pos];
            auto* i = blk.begin + vec_pos;
            if(i >= blk.curr) return false;
            if(i == blk.curr - 1)
            {
                bool const was_null = i->type == ::phy_engine::model::model_type::null;
                if(was_null) --blk.num_of_null_model;
                i->~model_base();
                --blk.curr;
                return !was_null;
            }
            if(i->type == ::phy_engine::model::model_type::null) return false;
            ++blk.num_of_null_model;
            i->clear();
            return true;
        }
        inline bool delete_model(netlist& nl, model_pos pos) noexcept { return delete_model(nl, pos.vec_pos, pos.chunk_pos); }

        inline ::phy_engine::model::node_t& create_node(netlist& nl)
        {
            if(nl.nodes.empty() || nl.nodes.back().size() == details::netlist_node_block::chunk_module_size) nl.nodes.emplace_back();
            auto& blk = nl.nodes.back();
            ::new(blk.curr)::phy_engine::model::node_t{};
            return *(blk.curr++);
        }

        // netlist/operation.h:167-205
        inline bool add_to_node([[maybe_unused]] netlist const& nl, ::phy_engine::model::model_base& model, ::std::size_t n1, ::phy_engine::model::node_t& node) noexcept
        {
            auto pw = model.ptr->generate_pin_view();
            if(n1 >= pw.size) return false;
            auto& p = pw.pins[n1];
            p.nodes = &node;
            node.pins.insert(&p);
            if(model.ptr->get_device_type() != ::phy_engine::model::model_device_type::digital) ++node.num_of_analog_node;
            return true;
        }
        inline bool add_to_node(netlist& nl, model_pos mp, ::std::size_t n1, ::phy_engine::model::node_t& node) noexcept
        {
            auto* m = get_model(nl, mp);
            return m ? add_to_node(nl, *m, n1, node) : false;
        }
        inline bool remove_from_node([[maybe_unused]] netlist const& nl, ::phy_engine::model::model_base& model, ::std::size_t n1, ::phy_engine::model::node_t& node) noexcept
        {
            auto pw = model.ptr->generate_pin_view();
            if(n1 >= pw.size) return false;
            auto& p = pw.pins[n1];
            p.nodes = nullptr;
            node.pins.erase(&p);
            if(model.ptr->get_device_type() != ::phy_engine::model::model_device_type::digital) --node.num_of_analog_node;
            return true;
        }
        inline void delete_node([[maybe_unused]] netlist const& nl, ::phy_engine::model::node_t& node) noexcept { node.clear(); }

        // netlist/operation.h:251-259 -- including its quirk: the survivor's num_of_analog_node is NOT increased
        // (SURVEY.md appendix A.7); analog indexing below therefore also accepts nodes that hold analog pins.
        inline void merge_node([[maybe_unused]] netlist const& nl, ::phy_engine::model::node_t& node, ::phy_engine::model::node_t& other_node) noexcept
        {
            for(auto* i: other_node.pins)
            {
                node.pins.insert(i);
                i->nodes = &node;
            }
            other_node.destroy();
        }

        template <bool check = false>
        inline ::std::size_t get_num_of_model(netlist const& nl) noexcept
        {
            ::std::size_t res{};
            for(auto const& b: nl.models)
                for(auto* p = b.begin; p != b.curr; ++p)
                    if(!check || p->type != ::phy_engine::model::model_type::null) ++res;
            return res;
        }
    }  // namespace netlist

    // ---- circuits/circuit.h:60-1528
    struct circult
    {
        environment env{};
        ::phy_engine::netlist::netlist nl{};
        analyze_type at{};
        ::phy_engine::analyzer::analyzer_storage_t analyzer_setting{};

        bool has_prepare{};
        ::std::size_t node_counter{};
        ::std::size_t branch_counter{};
        ::fast_io::vector<::phy_engine::model::node_t*> size_t_to_node_p{};
        ::fast_io::vector<::phy_engine::model::branch*> size_t_to_branch_p{};
        ::phy_engine::digital::digital_node_update_table digital_update_tables{};
        ::std::vector<::phy_engine::digital::need_operate_analog_node_t> digital_out{};
        ::std::vector<::phy_engine::model::model_base*> before_all_clk_digital_model{};
        ::std::vector<::phy_engine::model::model_base*> after_all_clk_digital_model{};
        double tr_duration{};
        double last_step{};
        ::std::string last_error{};
        pe_hip_run_stats last_stats{};
        // circuit.h:100: the assembled system.  Here: after an analysis of a circuit of up to `mna_mirror_rows` rows, a host copy of
        // the last system the device assembled (built-in models and host-stamped ones together), for inspection only.
        ::phy_engine::MNA::MNA mna{};
        ::std::size_t mna_mirror_rows{2048};
        // device-resident state handed over by pe_nl_fileformat::load (key runtime/pe_hip_state), applied when the circuit is next
        // loaded onto the device; device_state() is what pe_nl_fileformat::save stores (empty: nothing resident)
        ::std::string pending_device_state{};
        // the netlist's node voltages / branch currents were changed from outside (a checkpoint applied by pe_nl_fileformat::load):
        // the next analyze() loads the circuit onto the device again and resumes from them (and from pending_device_state)
        void adopt_netlist_state() noexcept
        {
            loaded_ = false;
            has_prepare = false;
        }
        ::std::string device_state() const
        {
            ::std::string blob;
            ::std::size_t n = 0;
            if(!gpu_ || !loaded_ || pe_hip_checkpoint_size(gpu_, &n) != PE_HIP_OK || n == 0) return blob;
            blob.resize(n);
            if(pe_hip_checkpoint_save(gpu_, blob.data(), n) != PE_HIP_OK) blob.clear();
            return blob;
        }

        // circuit.h:63-68,115-121: the reference chooses between its CPU LU and the CUDA solver per solve.  Kept for source
        // compatibility; this engine has ONE solver (the device), so auto_select and force_cuda mean the same and force_cpu makes
        // analyze() fail loudly (there is no CPU numeric path to fall back to).
        enum class cuda_solve_policy : ::std::uint_fast8_t
        {
            auto_select,
            force_cpu,
            force_cuda
        };
        inline static constexpr ::std::size_t default_cuda_node_threshold{100000};
        cuda_solve_policy cuda_policy{cuda_solve_policy::auto_select};
        ::std::size_t cuda_node_threshold{default_cuda_node_threshold};
        constexpr void set_cuda_policy(cuda_solve_policy p) noexcept { cuda_policy = p; }
        constexpr void set_cuda_node_threshold(::std::size_t n) noexcept { cuda_node_threshold = n; }

        circult() = default;
        circult(circult const&) = delete;
        circult& operator=(circult const&) = delete;
        ~circult()
        {
            if(gpu_) pe_hip_destroy(gpu_);
        }

        environment& get_environment() noexcept { return env; }
        ::phy_engine::netlist::netlist& get_netlist() noexcept { return nl; }
        void set_analyze_type(analyze_type other) noexcept { at = other; }
        ::phy_engine::analyzer::analyzer_storage_t& get_analyze_setting() noexcept { return analyzer_setting; }
        pe_hip_engine* gpu_engine() noexcept { return gpu_; }

        // circuit.h:179-296.  PHY_ENGINE_PROFILE_SOLVE=1 (circuit.h:35-58,1359-1479) prints one "[profile]" line per analyze() to
        // stderr -- per call, not per solve_once: the whole Newton / time loop of a call runs on the device.
        bool analyze() noexcept
        {
            if(cuda_policy == cuda_solve_policy::force_cpu)
            {
                last_error = "cuda_solve_policy::force_cpu: the MI355X engine has no CPU solver";
                return false;
            }
            static bool const prof = []
            {
                char const* v = ::std::getenv("PHY_ENGINE_PROFILE_SOLVE");
                return v && (*v == '1' || *v == 'y' || *v == 'Y' || *v == 't' || *v == 'T');
            }();
            if(!prof) return analyze_impl();
            auto const t0 = ::std::chrono::steady_clock::now();
            last_stats = pe_hip_run_stats{};
            bool const ok = analyze_impl();
            double const total_ms = ::std::chrono::duration<double, ::std::milli>(::std::chrono::steady_clock::now() - t0).count();
            pe_hip_info info{};
            if(gpu_ && loaded_) (void)pe_hip_get_info(gpu_, &info);
            ::std::fprintf(stderr, "[profile] n=%zu nnz=%d analysis=%u steps=%lld newton_iters=%lld hip_total_ms=%.6g (launches=%d dominant_ms=%.6g) ok=%d total_ms=%.6g\n",
                           node_counter + branch_counter, info.nnz_a, static_cast<unsigned>(at), last_stats.steps, last_stats.newton_iters, last_stats.gpu_ms,
                           last_stats.n_launches, last_stats.dominant_ms, ok ? 1 : 0, total_ms);
            return ok;
        }

    private:
        bool analyze_impl() noexcept
        {
            switch(at)
            {
                case analyze_type::OP: [[fallthrough]];
                case analyze_type::DC:
                    if(!prepare()) return false;
                    return solve();
                case analyze_type::TR: return run_tr(false);
                case analyze_type::TROP: return run_tr(true);
                case analyze_type::AC:  // circuit.h:192-212: operating point first when a non-linear device needs a linearisation
                case analyze_type::ACOP:  // circuit.h:213-232: always
                {
                    if(!prepare()) return false;
                    ac_sweep_results.clear();
                    if(at == analyze_type::ACOP || has_nonlinear_device())
                    {
                        auto const saved = at;
                        at = analyze_type::OP;
                        bool const ok = solve();
                        at = saved;
                        if(!ok) return false;
                    }
                    return run_ac_analysis();
                }
                default: return false;
            }
        }

    public:
        // circuit.h:70-82, 173-177
        struct ac_sweep_point
        {
            double omega{};
            ::std::vector<::std::complex<double>> x{};  // node voltages ; branch currents
        };
        ::std::vector<ac_sweep_point> ac_sweep_results{};
        auto& get_ac_sweep_results() noexcept { return ac_sweep_results; }
        void clear_ac_sweep_results() noexcept { ac_sweep_results.clear(); }

        // circuit.h:433-445
        [[nodiscard]] bool has_nonlinear_device() const noexcept
        {
            for(auto const& blk: nl.models)
                for(auto* c = blk.begin; c != blk.curr; ++c)
                    if(c->type == ::phy_engine::model::model_type::normal && c->ptr &&
                       c->ptr->get_device_type() == ::phy_engine::model::model_device_type::non_linear)
                        return true;
            return false;
        }

        // circuit.h:376-388
        [[nodiscard]] ::std::vector<::std::complex<double>> capture_solution_vector() const
        {
            ::std::vector<::std::complex<double>> x(node_counter + branch_counter);
            for(auto const* n: size_t_to_node_p) x[n->node_index] = n->node_information.an.voltage;
            for(auto const* b: size_t_to_branch_p) x[node_counter + b->index] = b->current;
            return x;
        }

        // circuit.h:389-431: single point, linear or logarithmic sweep of omega; every point is one AC solve on the device
        [[nodiscard]] bool run_ac_analysis() noexcept
        {
            auto& ac = analyzer_setting.ac;
            using sweep_t = ::phy_engine::analyzer::AC::sweep_type;
            if(ac.sweep == sweep_t::single || ac.points <= 1) return solve_ac_point(ac.omega);
            ac_sweep_results.clear();
            if(ac.sweep == sweep_t::linear)
            {
                double const step = (ac.omega_stop - ac.omega_start) / static_cast<double>(ac.points - 1);
                for(::std::size_t i = 0; i < ac.points; ++i)
                {
                    ac.omega = ac.omega_start + step * static_cast<double>(i);
                    if(!solve_ac_point(ac.omega)) return false;
                    ac_sweep_results.push_back({ac.omega, capture_solution_vector()});
                }
                return true;
            }
            if(ac.sweep == sweep_t::log)
            {
                if(ac.omega_start <= 0.0 || ac.omega_stop <= 0.0) return false;
                double const ratio = ::std::pow(ac.omega_stop / ac.omega_start, 1.0 / static_cast<double>(ac.points - 1));
                double omega = ac.omega_start;
                for(::std::size_t i = 0; i < ac.points; ++i)
                {
                    ac.omega = omega;
                    if(!solve_ac_point(ac.omega)) return false;
                    ac_sweep_results.push_back({ac.omega, capture_solution_vector()});
                    omega *= ratio;
                }
                return true;
            }
            return solve_ac_point(ac.omega);
        }

        // one AC solve (solve_once with iterate_ac): phasors scattered into the nodes / branches as complex values
        bool solve_ac_point(double omega) noexcept
        {
            if(!gpu_ || !loaded_) return false;
            ::std::size_t const rows = node_counter + branch_counter;
            if(!rows) return true;
            if(pe_hip_analyze_ac(gpu_, omega, &last_stats) != PE_HIP_OK) return gpu_fail();
            ::std::vector<double> re(rows), im(rows);
            if(pe_hip_get_solution_ac(gpu_, 0, 1, re.data(), im.data()) != PE_HIP_OK) return gpu_fail();
            for(auto* n: size_t_to_node_p) n->node_information.an.voltage = {re[n->node_index], im[n->node_index]};
            nl.ground_node.node_information.an.voltage = {};
            for(auto* b: size_t_to_branch_p) b->current = {re[node_counter + b->index], im[node_counter + b->index]};
            return true;
        }

        // circuit.h:298-354: one digital tick.  Runs on the host by design (integer / enum event logic);
        // models that drive analog nodes are collected in digital_out and become ideal sources of the next analyze().
        void digital_clk() noexcept
        {
            using namespace ::phy_engine::model;
            digital_out.clear();
            for(auto* i: before_all_clk_digital_model)
            {
                auto const rt = i->ptr->update_digital_clk(digital_update_tables, tr_duration, digital_update_method_t::before_all_clk);
                if(rt.need_to_operate_analog_node) digital_out.push_back(rt);
            }
            if(!digital_update_tables.always_tables.empty())
                digital_update_tables.tables.insert(digital_update_tables.always_tables.begin(), digital_update_tables.always_tables.end());
            ::std::size_t budget{10'000'000};
            while(!digital_update_tables.tables.empty())
            {
                if(budget-- == 0) break;
                auto it = digital_update_tables.tables.begin();
                auto* node = *it;
                digital_update_tables.tables.erase(it);
                for(auto* p: node->pins)
                {
                    auto* m = p->model;
                    if(m && m->ptr && m->ptr->get_device_type() == model_device_type::digital)
                    {
                        auto const rt = m->ptr->update_digital_clk(digital_update_tables, tr_duration, digital_update_method_t::update_table);
                        if(rt.need_to_operate_analog_node) digital_out.push_back(rt);
                    }
                }
            }
            for(auto* i: after_all_clk_digital_model)
            {
                auto const rt = i->ptr->update_digital_clk(digital_update_tables, tr_duration, digital_update_method_t::after_all_clk);
                if(rt.need_to_operate_analog_node) digital_out.push_back(rt);
            }
        }

        // circuit.h:446-465
        void reset() noexcept
        {
            tr_duration = 0.0;
            last_step = 0.0;
            digital_out.clear();
            digital_update_tables.always_tables.clear();
            digital_update_tables.tables.clear();
            for(auto* n: size_t_to_node_p) n->node_information.an.voltage = {};
            for(auto* b: size_t_to_branch_p) b->current = {};
            node_counter = branch_counter = 0;
            size_t_to_node_p.clear();
            size_t_to_branch_p.clear();
            has_prepare = false;
            if(gpu_ && loaded_) (void)pe_hip_reset(gpu_);
        }

        // circuit.h:468-890: index nodes / branches and (re)build the device tables; loads the GPU engine when the
        // netlist (topology or parameters) differs from what is resident
        bool prepare() noexcept
        {
            using namespace ::phy_engine::model;
            digital_update_tables.always_tables.clear();
            digital_update_tables.tables.clear();
            node_counter = 0;
            size_t_to_node_p.clear();
            // pin -> model back pointers first (the node classification below looks at the models)
            for(auto& blk: nl.models)
                for(auto* c = blk.begin; c != blk.curr; ++c)
                    if(c->type == model_type::normal)
                    {
                        auto const pv = c->ptr->generate_pin_view();
                        for(::std::size_t i = 0; i < pv.size; ++i) pv.pins[i].model = c;
                    }
            for(auto& blk: nl.nodes)
                for(auto* c = blk.begin; c != blk.curr; ++c)
                {
                    ::std::size_t analog_pins = 0;  // counted from the pins: immune to the merge_node quirk
                    for(auto* p: c->pins)
                        if(p->model && p->model->ptr && p->model->ptr->get_device_type() != model_device_type::digital) ++analog_pins;
                    if(analog_pins == 0)
                    {
                        if(!c->pins.empty() && !has_prepare) c->node_information.dn.state = digital_node_statement_t::X;  // circuit.h:485-490
                        continue;
                    }
                    if(analog_pins != c->pins.size()) digital_update_tables.always_tables.emplace(c);  // hybrid (circuit.h:494-497)
                    size_t_to_node_p.push_back(c);
                    c->node_index = node_counter++;
                }
            nl.ground_node.node_index = SIZE_MAX;
            branch_counter = digital_out.size();  // digital drives own the leading branch rows (circuit.h:509)
            size_t_to_branch_p.clear();
            before_all_clk_digital_model.clear();
            after_all_clk_digital_model.clear();

            tables_ next{};
            ::std::vector<model_base*> next_overlay{};
            for(auto& blk: nl.models)
                for(auto* c = blk.begin; c != blk.curr; ++c)
                {
                    if(c->type != model_type::normal) continue;
                    auto const pv = c->ptr->generate_pin_view();
                    for(::std::size_t i = 0; i < pv.size; ++i) pv.pins[i].model = c;
                    auto const bv = c->ptr->generate_branch_view();
                    ::std::size_t const branch0 = branch_counter;
                    for(::std::size_t i = 0; i < bv.size; ++i)
                    {
                        size_t_to_branch_p.push_back(bv.branches + i);
                        bv.branches[i].index = branch_counter++;
                    }
                    // internal nodes of the model follow the netlist's nodes, in model order (circuit.h:533-540)
                    auto const iv = c->ptr->generate_internal_node_view();
                    for(::std::size_t i = 0; i < iv.size; ++i)
                    {
                        size_t_to_node_p.push_back(iv.nodes + i);
                        iv.nodes[i].node_index = node_counter++;
                    }
                    if(c->ptr->get_device_type() == model_device_type::digital)
                    {
                        auto const method = static_cast<unsigned>(c->ptr->get_digital_update_method());
                        if(method & static_cast<unsigned>(digital_update_method_t::before_all_clk)) before_all_clk_digital_model.push_back(c);
                        else if(method & static_cast<unsigned>(digital_update_method_t::after_all_clk))
                            after_all_clk_digital_model.push_back(c);
                        c->has_init = true;  // (circuit.h:585-590; the digital blocks have no init hook)
                        continue;  // event logic stays on the host
                    }
                    // circuit.h:595-635 (every analysis branch): global TNOM reaches a model that exposes "tnom" and still has the
                    // default 27 C; the environment temperature reaches every model (hook, or the "Temp" attribute fallback)
                    if(env.norm_temperature != 27.0)
                    {
                        ::std::size_t const idx{c->ptr->find_attribute(u8"tnom")};
                        if(idx != SIZE_MAX)
                        {
                            auto const cur = c->ptr->get_attribute(idx);
                            if(!(cur.type == variant_type::d && ::std::abs(cur.d - 27.0) > 1e-12))
                            {
                                variant v{};
                                v.d = env.norm_temperature;
                                v.type = variant_type::d;
                                (void)c->ptr->set_attribute(idx, v);
                            }
                        }
                    }
                    if(!c->ptr->load_temperature(env.temperature))
                    {
                        last_error = "load_temperature_define returned false";
                        return false;
                    }
                    gpu_table_rows rows{};
                    // (circuit.h:585-590: every model is initialised once; the models with a device table have no init hook -- the flag
                    //  is kept because it is part of the wrapper data a PE-NL container carries, host-stamped models: prepare_overlay)
                    if(c->ptr->has_gpu_table()) c->has_init = true;
                    if(!c->ptr->has_gpu_table() || !c->ptr->gpu_table(rows))
                    {
                        // no device table: a plug-in model with the reference's host hooks only -> host-stamp overlay
                        if(c->ptr->has_host_stamp())
                        {
                            next_overlay.push_back(c);
                            continue;
                        }
                        auto const nm = c->ptr->get_model_name();
                        last_error = "model '" + ::std::string(reinterpret_cast<char const*>(nm.data()), nm.size()) +
                                     "' has neither a gpu_table_define hook nor an iterate_*_define hook";
                        return false;
                    }
                    for(int r = 0; r < rows.count; ++r)
                    {
                        auto const& row = rows.row[r];
                        auto node_id = [&](int pin) -> int
                        {
                            auto* n = pv.pins[pin].nodes;
                            if(!n) return -1;
                            return n == &nl.ground_node ? 0 : static_cast<int>(n->node_index) + 1;
                        };
                        auto& t = next.kind[row.kind];
                        t.nodes.push_back(node_id(row.pin_a));
                        t.nodes.push_back(node_id(row.pin_b));
                        if(model::gpu_kind_pins(row.kind) >= 3) t.nodes.push_back(node_id(row.pin_c));
                        if(model::gpu_kind_pins(row.kind) >= 4) t.nodes.push_back(node_id(row.pin_d));
                        if(model::gpu_kind_pins(row.kind) == 5) t.nodes.push_back(node_id(row.pin_e));
                        if(row.branch >= 0) t.branch.push_back(static_cast<int>(branch0) + row.branch);
                        if(row.branch2 >= 0) t.branch.push_back(static_cast<int>(branch0) + row.branch2);
                        if(row.branch3 >= 0) t.branch.push_back(static_cast<int>(branch0) + row.branch3);
                        int const ncol = model::gpu_kind_ncol(row.kind);
                        for(int q = 0; q < ncol; ++q) t.params.push_back(row.params[q]);
                    }
                }
            next.n_nodes = static_cast<int>(node_counter);
            next.n_branches = static_cast<int>(branch_counter);
            for(auto const& d: digital_out)  // ideal sources of circuit.h:1015-1022
            {
                auto* n = d.need_to_operate_analog_node;
                next.drv_node.push_back(n == &nl.ground_node ? 0 : static_cast<int>(n->node_index) + 1);
                next.drv_volt.push_back(d.voltage);
            }

            if(!gpu_)
            {
                if(pe_hip_create(0, &gpu_) != PE_HIP_OK)
                {
                    last_error = pe_hip_last_error(nullptr);
                    return false;
                }
            }
            if(!prepare_overlay(::std::move(next_overlay), next)) return false;
            pe_hip_options o{};
            o.v_abstol = env.V_eps_max;
            o.v_reltol = env.V_epsr_max;
            o.i_abstol = env.I_eps_max;
            o.i_reltol = env.I_epsr_max;
            o.g_min = env.g_min;
            o.r_open = env.r_open;
            o.refactor_every_solve = 1;
            if(pe_hip_set_options(gpu_, &o) != PE_HIP_OK) return gpu_fail();

            if(!loaded_ || !next.same_topology(resident_))
            {
                if(pe_hip_set_digital_drives(gpu_, static_cast<int>(next.drv_node.size()), next.drv_node.data(), next.drv_volt.data()) != PE_HIP_OK)
                    return gpu_fail();
                ::std::vector<pe_hip_device_table> tabs;
                for(int k = 1; k <= PE_HIP_KIND_MAX; ++k)
                {
                    auto& t = next.kind[k];
                    if(t.nodes.empty()) continue;
                    tabs.push_back({k, static_cast<int>(t.nodes.size()) / model::gpu_kind_pins(k), t.nodes.data(), t.branch.empty() ? nullptr : t.branch.data(),
                                    t.params.data(), 0});
                }
                if(pe_hip_load_circuit(gpu_, next.n_nodes, next.n_branches, 1, static_cast<int>(tabs.size()), tabs.data()) != PE_HIP_OK) return gpu_fail();
                loaded_ = true;
                // resume from the state the netlist carries (circult keeps node voltages / branch currents in the netlist)
                ::std::vector<double> x(node_counter + branch_counter, 0.0);
                for(auto* n: size_t_to_node_p) x[n->node_index] = n->node_information.an.voltage.real();
                for(auto* b: size_t_to_branch_p) x[node_counter + b->index] = b->current.real();
                if(!x.empty() && pe_hip_set_solution(gpu_, 0, 1, x.data()) != PE_HIP_OK) return gpu_fail();
                if(pe_hip_set_time(gpu_, tr_duration, last_step) != PE_HIP_OK) return gpu_fail();
                // a PE-NL container written by this build carries the device-resident state of the circuit it saved (companion
                // histories, junction / relay state, counters): a transient resumed from it continues bit for bit.  A blob that does
                // not fit this circuit (another topology) is dropped -- the netlist state above is the fallback.
                if(!pending_device_state.empty())
                {
                    (void)pe_hip_checkpoint_load(gpu_, pending_device_state.data(), pending_device_state.size());
                    pending_device_state.clear();
                }
            }
            else
            {
                if(!next.drv_node.empty() && next.drv_volt != resident_.drv_volt &&
                   pe_hip_set_digital_drives(gpu_, static_cast<int>(next.drv_node.size()), next.drv_node.data(), next.drv_volt.data()) != PE_HIP_OK)
                    return gpu_fail();
                // same topology: push changed parameters only
                for(int k = 1; k <= PE_HIP_KIND_MAX; ++k)
                {
                    auto const& a = next.kind[k].params;
                    auto const& b = resident_.kind[k].params;
                    int const ncol = model::gpu_kind_ncol(k);
                    for(::std::size_t i = 0; i < a.size(); ++i)
                        if(a[i] != b[i])
                            if(pe_hip_update_param(gpu_, k, static_cast<int>(i / ncol), static_cast<int>(i % ncol), &a[i], 0) != PE_HIP_OK) return gpu_fail();
                }
            }
            resident_ = ::std::move(next);
            has_prepare = true;
            return true;
        }

        // circuit.h:892-985 (+ solve_once): one OP / DC / TROP point, Newton on the device
        bool solve() noexcept
        {
            if(!gpu_ || !loaded_) return false;
            int const mode = at == analyze_type::OP ? PE_HIP_MODE_OP : (at == analyze_type::TROP ? PE_HIP_MODE_TROP : PE_HIP_MODE_DC);
            int const rc = node_counter + branch_counter == 0 ? PE_HIP_OK : pe_hip_analyze_dc(gpu_, mode, &last_stats);
            scatter();
            mirror_mna();
            if(rc != PE_HIP_OK) return gpu_fail();
            // circuit.h:965-975: the operating point is saved by the non-linear models once the solve has converged
            for(auto* c: overlay_models_)
                if(!c->ptr->save_op()) return overlay_hook_failed(c, "save_op_define");
            return true;
        }

    private:
        struct table_
        {
            ::std::vector<int> nodes, branch;
            ::std::vector<double> params;
        };
        struct tables_
        {
            int n_nodes{}, n_branches{};
            table_ kind[PE_HIP_KIND_MAX + 1]{};
            ::std::vector<int> drv_node;
            ::std::vector<double> drv_volt;
            ::std::vector<int> ov_rows, ov_cols, ov_rhs;  // host-stamp overlay cells (absolute MNA indices)
            bool ov_nonlinear{};
            bool same_topology(tables_ const& o) const
            {
                if(n_nodes != o.n_nodes || n_branches != o.n_branches || drv_node != o.drv_node) return false;
                if(ov_rows != o.ov_rows || ov_cols != o.ov_cols || ov_rhs != o.ov_rhs || ov_nonlinear != o.ov_nonlinear) return false;
                for(int k = 1; k <= PE_HIP_KIND_MAX; ++k)
                    if(kind[k].nodes != o.kind[k].nodes || kind[k].branch != o.kind[k].branch || kind[k].params.size() != o.kind[k].params.size()) return false;
                return true;
            }
        };

        pe_hip_engine* gpu_{};
        bool loaded_{};
        tables_ resident_{};

        bool gpu_fail() noexcept
        {
            last_error = pe_hip_last_error(gpu_);
            return false;
        }

        // ---- host-stamp overlay (include/pe_hip.h: pe_hip_set_overlay) ---------------------------------------------------------
        // Plug-in models without a device table keep the reference's contract: their iterate_*_define hooks stamp into an MNA.
        // prepare() runs every stamp hook such a model has once to DISCOVER the cells it touches (the pattern only grows:
        // mna_keep_pattern_ready, circuit.h:993-1003), registers those cells with the engine, and from then on the engine calls
        // back once per Newton iteration: node voltages / branch currents of the current iterate are scattered into the netlist
        // (the hooks read them through their pins, as in the reference), the hooks stamp, the values go to the device.
        ::std::vector<::phy_engine::model::model_base*> overlay_models_{};
        ::phy_engine::MNA::MNA overlay_mna_{};
        ::std::vector<::std::pair<::std::size_t, ::std::size_t>> overlay_cells_{};  // (row, col), the order of the uploaded values
        ::std::vector<::std::size_t> overlay_rhs_{};
        bool overlay_failed_{};

        bool overlay_stamp(int mode, double t) noexcept
        {
            overlay_mna_.r_open = env.r_open > 0.0 ? env.r_open : 1e12;
            for(auto* c: overlay_models_)
            {
                bool ok{};
                switch(mode)
                {
                    case PE_HIP_MODE_OP: ok = c->ptr->iterate_op(overlay_mna_); break;
                    case PE_HIP_MODE_TR: ok = c->ptr->iterate_tr(overlay_mna_, t); break;
                    case PE_HIP_MODE_TROP: ok = c->ptr->iterate_trop(overlay_mna_); break;
                    default: ok = c->ptr->iterate_dc(overlay_mna_); break;
                }
                if(!ok) return false;
            }
            return true;
        }
        void scatter_from(double const* x) noexcept
        {
            for(auto* n: size_t_to_node_p) n->node_information.an.voltage = x[n->node_index];
            nl.ground_node.node_information.an.voltage = {};
            for(auto* b: size_t_to_branch_p) b->current = x[node_counter + b->index];
        }
        static int overlay_trampoline(void* user, int event, int mode, double t, double dt, double const* x, double* a_values, double* b_values) noexcept
        {
            auto& self = *static_cast<circult*>(user);
            self.scatter_from(x);
            if(event == PE_HIP_OVERLAY_AC)
            {
                // circuit.h:389-431: one solve_once per frequency point with the models' iterate_ac hooks (complex stamps); t = omega
                self.overlay_mna_.clear_values_keep_pattern();
                self.overlay_mna_.r_open = self.env.r_open > 0.0 ? self.env.r_open : 1e12;
                for(auto* c: self.overlay_models_)
                    if(!c->ptr->iterate_ac(self.overlay_mna_, t)) return 1;
                ::std::size_t cells{};
                for(auto const& row: self.overlay_mna_.A) cells += row.size();
                if(cells != self.overlay_cells_.size() || self.overlay_mna_.Z.size() != self.overlay_rhs_.size())
                {
                    self.overlay_failed_ = true;
                    return 2;
                }
                ::std::size_t const nc{self.overlay_cells_.size()}, nr{self.overlay_rhs_.size()};
                for(::std::size_t i = 0; i < nc; ++i)
                {
                    auto const v = self.overlay_mna_.A[self.overlay_cells_[i].first][self.overlay_cells_[i].second];
                    a_values[i] = v.real();
                    a_values[nc + i] = v.imag();
                }
                for(::std::size_t i = 0; i < nr; ++i)
                {
                    auto const v = self.overlay_mna_.Z[self.overlay_rhs_[i]];
                    b_values[i] = v.real();
                    b_values[nr + i] = v.imag();
                }
                return 0;
            }
            if(event == PE_HIP_OVERLAY_CONVERGED)
            {
                // circuit.h:950-963: every model may veto an iterate that passed the Newton test
                for(auto* c: self.overlay_models_)
                    if(!c->ptr->check_convergence()) return PE_HIP_OVERLAY_VETO;
                return 0;
            }
            if(event == PE_HIP_OVERLAY_STEP)
            {
                for(auto* c: self.overlay_models_)
                    if(!c->ptr->step_changed_tr(self.last_step, dt)) return 1;
                self.last_step = dt;  // update_tr_step sets it after EVERY step (circuit.h:363-374), not once per analyze()
                return 0;
            }
            self.overlay_mna_.clear_values_keep_pattern();
            if(!self.overlay_stamp(mode, t)) return 1;
            // a cell outside the registered pattern cannot be added mid-analysis: fail loudly (the next prepare() re-discovers)
            ::std::size_t cells{};
            for(auto const& row: self.overlay_mna_.A) cells += row.size();
            if(cells != self.overlay_cells_.size() || self.overlay_mna_.Z.size() != self.overlay_rhs_.size())
            {
                self.overlay_failed_ = true;
                return 2;
            }
            for(::std::size_t i = 0; i < self.overlay_cells_.size(); ++i)
                a_values[i] = self.overlay_mna_.A[self.overlay_cells_[i].first][self.overlay_cells_[i].second].real();
            for(::std::size_t i = 0; i < self.overlay_rhs_.size(); ++i) b_values[i] = self.overlay_mna_.Z[self.overlay_rhs_[i]].real();
            return 0;
        }
        // init / prepare hooks (circuit.h:560-640), discovery stamp, registration with the engine
        bool prepare_overlay(::std::vector<::phy_engine::model::model_base*> models, tables_& next) noexcept
        {
            bool const same_models = models == overlay_models_;
            overlay_models_ = ::std::move(models);
            if(overlay_models_.empty())
            {
                overlay_cells_.clear();
                overlay_rhs_.clear();
                return pe_hip_set_overlay(gpu_, 0, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr) == PE_HIP_OK || gpu_fail();
            }
            bool nonlinear{};
            for(auto* c: overlay_models_)
            {
                if(!c->has_init)
                {
                    if(!c->ptr->init_model()) return overlay_hook_failed(c, "init_define");
                    c->has_init = true;
                }
                if(!c->ptr->prepare_for(static_cast<int>(at))) return overlay_hook_failed(c, "prepare_*_define");
                nonlinear = nonlinear || c->ptr->get_device_type() == ::phy_engine::model::model_device_type::non_linear;
            }
            if(!same_models || overlay_failed_ || overlay_mna_.node_size != node_counter || overlay_mna_.branch_size != branch_counter)
            {
                overlay_mna_ = ::phy_engine::MNA::MNA{node_counter, branch_counter};
                overlay_failed_ = false;
            }
            // discovery: every stamp the models can produce for this analysis (TR also needs the DC-like stamp of the first point)
            int const modes[2] = {at == analyze_type::TR || at == analyze_type::TROP ? PE_HIP_MODE_TR : (at == analyze_type::OP ? PE_HIP_MODE_OP : PE_HIP_MODE_DC),
                                  at == analyze_type::TROP ? PE_HIP_MODE_TROP : PE_HIP_MODE_DC};
            for(int q = 1; q >= 0; --q)  // the analysis' own mode last: its values are the representative ones
            {
                overlay_mna_.clear_values_keep_pattern();
                if(!overlay_stamp(modes[q], tr_duration)) return overlay_hook_failed(nullptr, "iterate_*_define");
            }
            // A hook may stamp cells later that the discovery stamp did not produce (a companion that is zero before the first step, a
            // region of operation the start point is not in): close the pattern over every cell a model CAN reach -- all pairs of its
            // own rows (pins, internal nodes, branches).  Cells only the closure adds weigh next to nothing in the pivot matching.
            ::std::vector<::std::pair<::std::size_t, ::std::size_t>> discovered;
            for(::std::size_t r = 0; r < overlay_mna_.A.size(); ++r)
                for(auto const& [col, v]: overlay_mna_.A[r]) discovered.emplace_back(r, col);
            for(auto* c: overlay_models_)
            {
                ::std::vector<::std::size_t> own;
                auto const pv = c->ptr->generate_pin_view();
                for(::std::size_t i = 0; i < pv.size; ++i)
                    if(pv.pins[i].nodes && pv.pins[i].nodes->node_index != SIZE_MAX) own.push_back(pv.pins[i].nodes->node_index);
                auto const iv = c->ptr->generate_internal_node_view();
                for(::std::size_t i = 0; i < iv.size; ++i) own.push_back(iv.nodes[i].node_index);
                auto const bv = c->ptr->generate_branch_view();
                for(::std::size_t i = 0; i < bv.size; ++i) own.push_back(node_counter + bv.branches[i].index);
                for(auto const r: own)
                {
                    for(auto const cc: own) overlay_mna_.A_ref(r, cc) += 0.0;
                    overlay_mna_.Z_ref(r) += 0.0;
                }
            }
            overlay_cells_.clear();
            overlay_rhs_.clear();
            ::std::vector<double> rep;
            for(::std::size_t r = 0; r < overlay_mna_.A.size(); ++r)
                for(auto const& [col, v]: overlay_mna_.A[r])
                {
                    overlay_cells_.emplace_back(r, col);
                    next.ov_rows.push_back(static_cast<int>(r));
                    next.ov_cols.push_back(static_cast<int>(col));
                    bool const seen = ::std::find(discovered.begin(), discovered.end(), ::std::pair<::std::size_t, ::std::size_t>{r, col}) != discovered.end();
                    rep.push_back(::std::abs(v) > 0.0 ? ::std::abs(v) : (seen ? 1.0 : 1e-30));
                }
            for(auto const& [row, v]: overlay_mna_.Z)
            {
                overlay_rhs_.push_back(row);
                next.ov_rhs.push_back(static_cast<int>(row));
            }
            next.ov_nonlinear = nonlinear;
            if(pe_hip_set_overlay(gpu_, static_cast<int>(next.ov_rows.size()), next.ov_rows.data(), next.ov_cols.data(), rep.data(), static_cast<int>(next.ov_rhs.size()),
                                  next.ov_rhs.data(), nonlinear ? 1 : 0, &circult::overlay_trampoline, this) != PE_HIP_OK)
                return gpu_fail();
            return true;
        }
        bool overlay_hook_failed(::phy_engine::model::model_base* c, char const* hook) noexcept
        {
            last_error = ::std::string("host-stamped model");
            if(c)
            {
                auto const nm = c->ptr->get_model_name();
                last_error += " '" + ::std::string(reinterpret_cast<char const*>(nm.data()), nm.size()) + "'";
            }
            last_error += ::std::string(": ") + hook + " returned false";
            return false;
        }
        // host copy of the system the device assembled last (small circuits only; circuit.h:100 `mna`)
        void mirror_mna() noexcept
        {
            ::std::size_t const rows = node_counter + branch_counter;
            mna.clear();
            mna.resize(node_counter, branch_counter);
            if(!rows || rows > mna_mirror_rows || !gpu_ || !loaded_) return;
            pe_hip_info info{};
            if(pe_hip_get_info(gpu_, &info) != PE_HIP_OK) return;
            int const nnz{info.nnz_a};
            ::std::vector<int> rp(rows + 1), ci(static_cast<::std::size_t>(nnz));
            ::std::vector<double> va(static_cast<::std::size_t>(nnz)), rhs(rows);
            if(pe_hip_get_matrix(gpu_, 0, rp.data(), ci.data(), va.data(), rhs.data()) != PE_HIP_OK) return;
            for(::std::size_t r = 0; r < rows; ++r)
            {
                for(int e = rp[r]; e < rp[r + 1]; ++e) mna.A[r][static_cast<::std::size_t>(ci[e])] = va[static_cast<::std::size_t>(e)];
                if(rhs[r] != 0.0) mna.Z[r] = rhs[r];
            }
        }

        // circuit.h:1521-1523
        void scatter() noexcept
        {
            ::std::size_t const rows = node_counter + branch_counter;
            if(!rows) return;
            ::std::vector<double> x(rows);
            if(pe_hip_get_solution(gpu_, 0, 1, x.data()) != PE_HIP_OK) return;
            for(auto* n: size_t_to_node_p) n->node_information.an.voltage = x[n->node_index];
            nl.ground_node.node_information.an.voltage = {};
            for(auto* b: size_t_to_branch_p) b->current = x[node_counter + b->index];
        }

        // circuit.h:233-289: the step count is what the reference's floating-point loop bound yields
        bool run_tr(bool trop) noexcept
        {
            double const dt = analyzer_setting.tr.t_step;
            if(dt <= 0.0) return false;
            double const t_stop = analyzer_setting.tr.t_stop;
            if(!prepare()) return false;
            if(trop)
            {
                auto const saved = at;
                at = analyze_type::TROP;
                bool const ok = solve();
                at = saved;
                if(!ok) return false;
            }
            int n = 0;
            {
                double t = tr_duration;
                double const end_time = tr_duration + t_stop;
                for(; t < end_time; t = t + dt) ++n;
            }
            int const rc = (n == 0 || node_counter + branch_counter == 0) ? PE_HIP_OK : pe_hip_analyze_tr(gpu_, dt, n, &last_stats);
            scatter();
            mirror_mna();
            double t_now = tr_duration;
            long long steps = 0;
            if(node_counter + branch_counter != 0)
                (void)pe_hip_get_instance_state(gpu_, 0, 1, nullptr, &steps, nullptr, &t_now);
            else
                for(int i = 0; i < n; ++i) t_now = t_now + dt;
            tr_duration = t_now;  // on failure the engine has rolled the failing step back (circuit.h:249-253)
            last_step = dt;
            if(rc != PE_HIP_OK) return gpu_fail();
            return true;
        }
    };
}  // namespace phy_engine
