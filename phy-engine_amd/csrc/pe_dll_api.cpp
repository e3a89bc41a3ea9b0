// pe_dll_api.cpp -- Phy-Engine's FFI netlist loader + control subset (include/phy_engine_dll_api.h) on top of the C++
// host layer (phy-engine_amd/include/phy_engine) and therefore on the MI355X engine.  Behaviour restated from
// src/dll_main.cpp of the reference (cited per function); no reference code is used.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numbers>
#include <numeric>
#include <string>
#include <vector>

#include <phy_engine/phy_engine.h>

#include "../../include/phy_engine_dll_api.h"

namespace
{
    namespace pe = ::phy_engine;
    thread_local std::string g_err;

    void set_err(std::string s) { g_err = std::move(s); }
}  // namespace
// (used by pe_dll_stubs.cpp: the refusing entry points report through the same thread-local message)
void pe_dll_set_error(std::string const& s) { g_err = s; }
namespace
{

    // element code -> model; properties are consumed positionally (src/dll_main.cpp:1707-1830, appendix C of SURVEY.md)
    bool add_element(pe::netlist::netlist& nl, int code, double const*& prop, pe::netlist::add_model_retstr& out)
    {
        using namespace pe::model;
        auto take = [&]() { return *prop++; };
        switch(code)
        {
            case 1: out = add_model(nl, resistance{.r = *prop++}); return true;
            case 2: out = add_model(nl, capacitor{.m_kZimag = *prop++}); return true;
            case 3: out = add_model(nl, inductor{.m_kZimag = *prop++}); return true;
            case 4: out = add_model(nl, VDC{.V = *prop++}); return true;
            case 5:
            {
                // VAC{Vp, f[Hz], phase[deg]} -> omega = 2*pi*f, phase in radians (dll_main.cpp:1729-1741)
                double const vp = *prop++, hz = *prop++, deg = *prop++;
                out = add_model(nl, VAC{.m_Vp = vp, .m_omega = 2.0 * std::numbers::pi * hz, .m_phase = deg * std::numbers::pi / 180.0});
                return true;
            }
            case 6: out = add_model(nl, IDC{.I = *prop++}); return true;
            case 13:
            {
                PN_junction d{};
                d.Is = *prop++;
                d.N = *prop++;
                d.Isr = *prop++;
                d.Nr = *prop++;
                d.Temp = *prop++;
                d.Ibv = *prop++;
                d.Bv = *prop++;
                d.Bv_set = (*prop++ != 0.0);
                d.Area = *prop++;
                out = add_model(nl, d);
                return true;
            }
            case 54: out = add_model(nl, full_bridge_rectifier{}); return true;
            // ---- SURVEY.md 8f rank 1 (dll_api.h:60-97): properties consumed positionally, in the order of the header comment
            case 7:
            {
                // IAC{Ip, f[Hz], phase[deg]}, same unit conversion as VAC
                double const ip = take(), hz = take(), deg = take();
                out = add_model(nl, IAC{.m_Ip = ip, .m_omega = 2.0 * std::numbers::pi * hz, .m_phase = deg * std::numbers::pi / 180.0});
                return true;
            }
            case 8: out = add_model(nl, VCCS{.m_g = take()}); return true;
            case 9: out = add_model(nl, VCVS{.m_mu = take()}); return true;
            case 10: out = add_model(nl, CCCS{.m_alpha = take()}); return true;
            case 11: out = add_model(nl, CCVS{.m_r = take()}); return true;
            case 12: out = add_model(nl, single_pole_switch{.cut_through = take() != 0.0}); return true;
            case 14: out = add_model(nl, transformer{.n = take()}); return true;
            case 15:
            {
                coupled_inductors kl{};
                for(double* f: {&kl.L1, &kl.L2, &kl.k}) *f = take();
                out = add_model(nl, kl);
                return true;
            }
            case 16: out = add_model(nl, transformer_center_tap{.n_total = take()}); return true;
            case 17: out = add_model(nl, op_amp{.mu = take()}); return true;
            case 18:
            {
                relay r{};
                r.Von = take();
                r.Voff = take();
                out = add_model(nl, r);
                return true;
            }
            case 20:
            {
                sawtooth_gen g{};
                for(double* f: {&g.Vh, &g.Vl, &g.freq, &g.phase}) *f = take();
                out = add_model(nl, g);
                return true;
            }
            case 21:
            {
                square_gen g{};
                for(double* f: {&g.Vh, &g.Vl, &g.freq, &g.duty, &g.phase}) *f = take();
                out = add_model(nl, g);
                return true;
            }
            case 22:
            {
                pulse_gen g{};
                for(double* f: {&g.Vh, &g.Vl, &g.freq, &g.duty, &g.phase, &g.tr, &g.tf}) *f = take();
                out = add_model(nl, g);
                return true;
            }
            case 23:
            {
                triangle_gen g{};
                for(double* f: {&g.Vh, &g.Vl, &g.freq, &g.phase}) *f = take();
                out = add_model(nl, g);
                return true;
            }
            // ---- mixed-signal subset (19, 200, 201, 204, 205) + the other two-input gates and the buffer
            case 19:
            {
                comparator c{};
                c.Ll = take();
                c.Hl = take();
                out = add_model(nl, c);
                return true;
            }
            case 200:
            {
                // INPUT{state: 0 L, 1 H, 2 X, 3 Z}
                int const st = static_cast<int>(take());
                INPUT in{};
                in.outputA = st == 0 ? digital_node_statement_t::L : st == 1 ? digital_node_statement_t::H : st == 3 ? digital_node_statement_t::Z : digital_node_statement_t::X;
                out = add_model(nl, in);
                return true;
            }
            case 201: out = add_model(nl, OUTPUT{}); return true;
            case 202: out = add_model(nl, OR{}); return true;
            case 203: out = add_model(nl, YES{}); return true;
            case 204: out = add_model(nl, AND{}); return true;
            case 205: out = add_model(nl, NOT{}); return true;
            case 206: out = add_model(nl, XOR{}); return true;
            case 207: out = add_model(nl, XNOR{}); return true;
            case 208: out = add_model(nl, NAND{}); return true;
            case 209: out = add_model(nl, NOR{}); return true;
            // ---- digital blocks (dll_api.h:110-124): host event queue only
            case 210: out = add_model(nl, TRI{}); return true;
            case 211: out = add_model(nl, IMP{}); return true;
            case 212: out = add_model(nl, NIMP{}); return true;
            case 220: out = add_model(nl, HALF_ADDER{}); return true;
            case 221: out = add_model(nl, FULL_ADDER{}); return true;
            case 222: out = add_model(nl, HALF_SUB{}); return true;
            case 223: out = add_model(nl, FULL_SUB{}); return true;
            case 224: out = add_model(nl, MUL2{}); return true;
            case 225: out = add_model(nl, DFF{}); return true;
            case 226: out = add_model(nl, TFF{}); return true;
            case 227: out = add_model(nl, T_BAR_FF{}); return true;
            case 228: out = add_model(nl, JKFF{}); return true;
            case 229:
            {
                // COUNTER4{init_value 0..15}
                double const v = take();
                COUNTER4 ctr{};
                ctr.value = static_cast<::std::uint8_t>(v < 0.0 ? 0u : (v > 15.0 ? 15u : static_cast<unsigned>(v)));
                out = add_model(nl, ctr);
                return true;
            }
            case 230:
            {
                // RANDOM_GENERATOR4{init_state 0..15}
                double const v = take();
                RANDOM_GENERATOR4 g{};
                g.state = static_cast<::std::uint8_t>(v < 0.0 ? 0u : (v > 15.0 ? 15u : static_cast<unsigned>(v)));
                out = add_model(nl, g);
                return true;
            }
            case 231:
            {
                // EIGHT_BIT_INPUT{value 0..255}
                double const v = take();
                EIGHT_BIT_INPUT g{};
                g.value = static_cast<::std::uint8_t>(v < 0.0 ? 0u : (v > 255.0 ? 255u : static_cast<unsigned>(v)));
                out = add_model(nl, g);
                return true;
            }
            case 232: out = add_model(nl, EIGHT_BIT_DISPLAY{}); return true;
            case 233:
            {
                // SCHMITT_TRIGGER{Vth_low, Vth_high, inverted, Ll, Hl}
                SCHMITT_TRIGGER g{};
                g.Vth_low = take();
                g.Vth_high = take();
                g.inverted = take() != 0.0;
                g.Ll = take();
                g.Hl = take();
                out = add_model(nl, g);
                return true;
            }
            case 50:
            case 51:
            {
                // BJT{Is, N, BetaF, Temp, Area}
                double v[5];
                for(double& q: v) q = take();
                if(code == 50) out = add_model(nl, BJT_NPN{.Is = v[0], .N = v[1], .BetaF = v[2], .Temp = v[3], .Area = v[4]});
                else
                    out = add_model(nl, BJT_PNP{.Is = v[0], .N = v[1], .BetaF = v[2], .Temp = v[3], .Area = v[4]});
                return true;
            }
            case 52:
            case 53:
            {
                // level-1 MOSFET{Kp, lambda, Vth}
                double const kp = take(), lam = take(), vth = take();
                if(code == 52) out = add_model(nl, nmosfet{.Kp = kp, .lambda = lam, .Vth = vth});
                else
                    out = add_model(nl, pmosfet{.Kp = kp, .lambda = lam, .Vth = vth});
                return true;
            }
            default: return false;
        }
    }
}  // namespace

extern "C" int circuit_sample_layout(void*, size_t*, size_t*, size_t, size_t*, size_t*, size_t*);

namespace
{
    template <typename D>
    int sample_impl(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord, double* current,
                    size_t* current_ord, D* digital, size_t* digital_ord)
    {
        if(!circuit_ptr || !vec_pos || !chunk_pos || !voltage || !voltage_ord || !current || !current_ord || !digital || !digital_ord) return 1;
        if(circuit_sample_layout(circuit_ptr, vec_pos, chunk_pos, comp_size, voltage_ord, current_ord, digital_ord) != 0) return 1;
        auto& nl = static_cast<pe::circult*>(circuit_ptr)->get_netlist();
        for(size_t i = 0; i < comp_size; ++i)
        {
            auto* m = get_model(nl, vec_pos[i], chunk_pos[i]);
            if(!m || !m->ptr) continue;
            auto const pv = m->ptr->generate_pin_view();
            auto const bv = m->ptr->generate_branch_view();
            for(size_t j = 0; j < pv.size; ++j)
            {
                auto const* node = pv.pins[j].nodes;
                voltage[voltage_ord[i] + j] = node ? node->node_information.an.voltage.real() : 0.0;
                bool v = false;
                if(node && node->num_of_analog_node == 0) v = node->node_information.dn.state == pe::model::digital_node_statement_t::true_state;
                digital[digital_ord[i] + j] = static_cast<D>(v);
            }
            for(size_t j = 0; j < bv.size; ++j) current[current_ord[i] + j] = bv.branches[j].current.real();
        }
        return 0;
    }
}  // namespace

extern "C" {

char const* phy_engine_last_error(void) { return g_err.c_str(); }
void phy_engine_clear_error(void) { g_err.clear(); }

// src/dll_main.cpp:2492-2602 + build_netlist_from_wires :1530-1700
void* create_circuit(int* elements, size_t ele_size, int* wires, size_t wires_size, double* properties, size_t** vec_pos, size_t** chunk_pos, size_t* comp_size)
{
    phy_engine_clear_error();
    if(!elements || !vec_pos || !chunk_pos || !comp_size || (wires_size && !wires))
    {
        set_err("create_circuit: null argument");
        return nullptr;
    }
    if(wires_size % 4 != 0)
    {
        set_err("create_circuit: wires_size must be a multiple of 4");
        return nullptr;
    }
    void* mem = std::malloc(sizeof(pe::circult));
    if(!mem)
    {
        set_err("create_circuit: out of memory");
        return nullptr;
    }
    auto* c = new(mem) pe::circult{};
    c->set_analyze_type(pe::analyze_type::TR);  // defaults of dll_main.cpp:2531-2536
    c->get_analyze_setting().tr.t_step = 1e-6;
    c->get_analyze_setting().tr.t_stop = 1e-6;
    auto& nl = c->get_netlist();

    std::vector<pe::model::model_base*> model_of(ele_size, nullptr);
    std::vector<pe::netlist::model_pos> pos;
    double const* prop = properties;
    for(size_t i = 0; i < ele_size; ++i)
    {
        if(elements[i] == 0) continue;  // ground placeholder
        pe::netlist::add_model_retstr r{};
        if(!add_element(nl, elements[i], prop, r))
        {
            set_err("create_circuit: element code " + std::to_string(elements[i]) + " is not supported by the MI355X loader subset");
            c->~circult();
            std::free(mem);
            return nullptr;
        }
        model_of[i] = r.mod;
        pos.push_back(r.mod_pos);
    }

    // union-find over (element, pin) slots; a net that touches a code-0 element is the ground net
    std::vector<size_t> first_slot(ele_size + 1, 0);
    for(size_t i = 0; i < ele_size; ++i)
    {
        size_t const np = model_of[i] ? model_of[i]->ptr->generate_pin_view().size : 1;  // the ground placeholder has one terminal
        first_slot[i + 1] = first_slot[i] + np;
    }
    std::vector<size_t> parent(first_slot[ele_size]);
    std::iota(parent.begin(), parent.end(), size_t{0});
    auto find = [&](size_t a)
    {
        while(parent[a] != a)
        {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    std::vector<char> wired(parent.size(), 0);
    for(size_t w = 0; w + 3 < wires_size; w += 4)
    {
        long const e1 = wires[w], p1 = wires[w + 1], e2 = wires[w + 2], p2 = wires[w + 3];
        if(e1 < 0 || e2 < 0 || static_cast<size_t>(e1) >= ele_size || static_cast<size_t>(e2) >= ele_size || p1 < 0 || p2 < 0) continue;  // skipped silently
        size_t const n1 = first_slot[e1 + 1] - first_slot[e1], n2 = first_slot[e2 + 1] - first_slot[e2];
        size_t const q1 = elements[e1] == 0 ? 0 : static_cast<size_t>(p1), q2 = elements[e2] == 0 ? 0 : static_cast<size_t>(p2);
        if(q1 >= n1 || q2 >= n2) continue;
        size_t const a = find(first_slot[e1] + q1), b = find(first_slot[e2] + q2);
        wired[first_slot[e1] + q1] = wired[first_slot[e2] + q2] = 1;
        if(a != b) parent[b] = a;
    }
    std::vector<char> is_ground(parent.size(), 0);
    for(size_t i = 0; i < ele_size; ++i)
        if(elements[i] == 0) is_ground[find(first_slot[i])] = 1;
    std::vector<pe::model::node_t*> node_of(parent.size(), nullptr);
    for(size_t i = 0; i < ele_size; ++i)
    {
        if(!model_of[i]) continue;
        size_t const np = first_slot[i + 1] - first_slot[i];
        for(size_t p = 0; p < np; ++p)
        {
            size_t const slot = first_slot[i] + p;
            if(!wired[slot]) continue;  // unconnected pins stay nullptr
            size_t const root = find(slot);
            pe::model::node_t* n;
            if(is_ground[root]) n = &nl.ground_node;
            else
            {
                if(!node_of[root]) node_of[root] = &create_node(nl);
                n = node_of[root];
            }
            add_to_node(nl, *model_of[i], p, *n);
        }
    }

    *comp_size = pos.size();
    *vec_pos = static_cast<size_t*>(std::malloc(sizeof(size_t) * (pos.size() ? pos.size() : 1)));
    *chunk_pos = static_cast<size_t*>(std::malloc(sizeof(size_t) * (pos.size() ? pos.size() : 1)));
    if(!*vec_pos || !*chunk_pos)
    {
        std::free(*vec_pos);
        std::free(*chunk_pos);
        *vec_pos = *chunk_pos = nullptr;
        c->~circult();
        std::free(mem);
        set_err("create_circuit: out of memory");
        return nullptr;
    }
    for(size_t i = 0; i < pos.size(); ++i)
    {
        (*vec_pos)[i] = pos[i].vec_pos;
        (*chunk_pos)[i] = pos[i].chunk_pos;
    }
    return c;
}

// src/dll_main.cpp:2861-2881
// src/dll_main.cpp create_circuit_ex: the same netlist builder; Verilog elements are the only thing the string table serves
void* create_circuit_ex(int* elements, size_t ele_size, int* wires, size_t wires_size, double* properties, char const* const*, size_t const*, size_t,
                        size_t const*, size_t const*, size_t** vec_pos, size_t** chunk_pos, size_t* comp_size)
{
    if(elements)
        for(size_t i = 0; i < ele_size; ++i)
            if(elements[i] == 300 || elements[i] == 301)
            {
                phy_engine_clear_error();
                set_err("create_circuit_ex: Verilog elements (code " + std::to_string(elements[i]) + ") are not supported by the MI355X loader");
                return nullptr;
            }
    return create_circuit(elements, ele_size, wires, wires_size, properties, vec_pos, chunk_pos, comp_size);
}

void destroy_circuit(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos)
{
    if(circuit_ptr)
    {
        static_cast<pe::circult*>(circuit_ptr)->~circult();
        std::free(circuit_ptr);
    }
    std::free(vec_pos);
    std::free(chunk_pos);
}

int circuit_set_analyze_type(void* circuit_ptr, uint32_t v)
{
    if(!circuit_ptr || v > 5u)
    {
        set_err("circuit_set_analyze_type: bad argument");
        return 1;
    }
    static_cast<pe::circult*>(circuit_ptr)->set_analyze_type(static_cast<pe::analyze_type>(v));
    return 0;
}

// dll_api.h:175: single-point AC at omega [rad/s]
int circuit_set_ac_omega(void* circuit_ptr, double omega)
{
    if(!circuit_ptr) return 1;
    auto& ac = static_cast<pe::circult*>(circuit_ptr)->get_analyze_setting().ac;
    ac.sweep = pe::analyzer::AC::sweep_type::single;
    ac.omega = omega;
    return 0;
}

int circuit_set_tr(void* circuit_ptr, double t_step, double t_stop)
{
    if(!circuit_ptr)
    {
        set_err("circuit_set_tr: null circuit");
        return 1;
    }
    auto& s = static_cast<pe::circult*>(circuit_ptr)->get_analyze_setting();
    s.tr.t_step = t_step;
    s.tr.t_stop = t_stop;
    return 0;
}

int circuit_set_temperature(void* circuit_ptr, double temp_c)
{
    if(!circuit_ptr) return 1;
    static_cast<pe::circult*>(circuit_ptr)->env.temperature = temp_c;
    return 0;
}

int circuit_set_tnom(void* circuit_ptr, double tnom_c)
{
    if(!circuit_ptr) return 1;
    static_cast<pe::circult*>(circuit_ptr)->env.norm_temperature = tnom_c;
    return 0;
}

int circuit_set_model_double_by_name(void* circuit_ptr, size_t vec_pos, size_t chunk_pos, char const* name, size_t name_size, double value)
{
    if(!circuit_ptr || !name)
    {
        set_err("circuit_set_model_double_by_name: null argument");
        return 1;
    }
    auto* c = static_cast<pe::circult*>(circuit_ptr);
    auto* m = get_model(c->get_netlist(), vec_pos, chunk_pos);
    if(!m || !m->ptr)
    {
        set_err("circuit_set_model_double_by_name: no such model");
        return 1;
    }
    auto ieq = [](char a, char b)
    {
        auto lo = [](char x) { return (x >= 'A' && x <= 'Z') ? static_cast<char>(x + ('a' - 'A')) : x; };
        return lo(a) == lo(b);
    };
    for(size_t idx = 0; idx < 512; ++idx)
    {
        auto const n = m->ptr->get_attribute_name(idx);
        if(n.size() != name_size) continue;
        bool same = true;
        for(size_t k = 0; k < name_size && same; ++k) same = ieq(static_cast<char>(n[k]), name[k]);
        if(!same) continue;
        pe::model::variant v{};
        v.d = value;
        v.type = pe::model::variant_type::d;
        if(m->ptr->set_attribute(idx, v)) return 0;
    }
    set_err("circuit_set_model_double_by_name: attribute not found");
    return 1;
}

// src/dll_main.cpp:2254-2259
int circuit_analyze(void* circuit_ptr)
{
    if(!circuit_ptr) return 1;
    auto* c = static_cast<pe::circult*>(circuit_ptr);
    if(c->analyze()) return 0;
    set_err("circuit_analyze: " + c->last_error);
    return 1;
}

int circuit_digital_clk(void* circuit_ptr)
{
    if(!circuit_ptr) return 1;
    static_cast<pe::circult*>(circuit_ptr)->digital_clk();  // host event queue (circuit.h:348-354); analog side untouched
    return 0;
}

// src/dll_main.cpp:2269-2305
int circuit_sample_layout(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, size_t* voltage_ord, size_t* current_ord, size_t* digital_ord)
{
    if(!circuit_ptr || !vec_pos || !chunk_pos || !voltage_ord || !current_ord || !digital_ord) return 1;
    auto& nl = static_cast<pe::circult*>(circuit_ptr)->get_netlist();
    voltage_ord[0] = current_ord[0] = digital_ord[0] = 0;
    for(size_t i = 0; i < comp_size; ++i)
    {
        auto* m = get_model(nl, vec_pos[i], chunk_pos[i]);
        size_t const np = (m && m->ptr) ? m->ptr->generate_pin_view().size : 0;
        size_t const nb = (m && m->ptr) ? m->ptr->generate_branch_view().size : 0;
        voltage_ord[i + 1] = voltage_ord[i] + np;
        current_ord[i + 1] = current_ord[i] + nb;
        digital_ord[i + 1] = digital_ord[i] + np;
    }
    return 0;
}

int circuit_sample(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord, double* current,
                   size_t* current_ord, bool* digital, size_t* digital_ord)
{
    return sample_impl(circuit_ptr, vec_pos, chunk_pos, comp_size, voltage, voltage_ord, current, current_ord, digital, digital_ord);
}

int circuit_sample_u8(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord, double* current,
                      size_t* current_ord, uint8_t* digital, size_t* digital_ord)
{
    return sample_impl(circuit_ptr, vec_pos, chunk_pos, comp_size, voltage, voltage_ord, current, current_ord, digital, digital_ord);
}

// src/dll_main.cpp:2883-2934: apply (element, attribute index, value) updates, analyze, sample.
// Returns 1 when the analysis fails and -- like the reference -- 0 when a required pointer is null.
// dll_api.h:222-234: like circuit_sample_u8 but every digital pin reports its 4-state value (0 L, 1 H, 2 X, 3 Z; X for analog pins)
int circuit_sample_digital_state_u8(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord,
                                    double* current, size_t* current_ord, uint8_t* digital, size_t* digital_ord)
{
    if(sample_impl<uint8_t>(circuit_ptr, vec_pos, chunk_pos, comp_size, voltage, voltage_ord, current, current_ord, digital, digital_ord) != 0) return 1;
    auto& nl = static_cast<pe::circult*>(circuit_ptr)->get_netlist();
    for(size_t i = 0; i < comp_size; ++i)
    {
        auto* m = get_model(nl, vec_pos[i], chunk_pos[i]);
        if(!m || !m->ptr) continue;
        auto const pv = m->ptr->generate_pin_view();
        for(size_t j = 0; j < pv.size; ++j)
        {
            auto const* node = pv.pins[j].nodes;
            uint8_t v = static_cast<uint8_t>(pe::model::digital_node_statement_t::X);
            if(node && node->num_of_analog_node == 0) v = static_cast<uint8_t>(node->node_information.dn.state);
            digital[digital_ord[i] + j] = v;
        }
    }
    return 0;
}

// dll_api.h:234: set a digital attribute (e.g. INPUT's value) of one component: 0 L, 1 H, 2 X, 3 Z
int circuit_set_model_digital(void* circuit_ptr, size_t vec_pos, size_t chunk_pos, size_t attribute_index, uint8_t state)
{
    if(!circuit_ptr) return 1;
    auto& nl = static_cast<pe::circult*>(circuit_ptr)->get_netlist();
    auto* m = get_model(nl, vec_pos, chunk_pos);
    if(!m || !m->ptr) return 2;
    using st = pe::model::digital_node_statement_t;
    pe::model::variant vi{};
    vi.digital = state == 0 ? st::L : state == 1 ? st::H : state == 3 ? st::Z : st::X;
    vi.type = pe::model::variant_type::digital;
    return m->ptr->set_attribute(attribute_index, vi) ? 0 : 3;
}

void phy_engine_string_free(char* s) { std::free(s); }

int analyze_circuit(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, int* changed_ele, size_t* changed_ind, double* changed_prop,
                    size_t prop_size, double* voltage, size_t* voltage_ord, double* current, size_t* current_ord, bool* digital, size_t* digital_ord)
{
    if(!(circuit_ptr && vec_pos && chunk_pos && voltage && voltage_ord && current && current_ord && digital && digital_ord)) return 0;
    auto* c = static_cast<pe::circult*>(circuit_ptr);
    auto& nl = c->get_netlist();
    for(size_t i = 0; i < prop_size; ++i)
    {
        auto* m = get_model(nl, vec_pos[changed_ele[i]], chunk_pos[changed_ele[i]]);
        if(!m || !m->ptr) continue;
        pe::model::variant v{};
        v.d = changed_prop[i];
        v.type = pe::model::variant_type::d;
        if(m->ptr->set_attribute(changed_ind[i], v)) continue;
        pe::model::variant bvar{};
        bvar.boolean = changed_prop[i] != 0.0;
        bvar.type = pe::model::variant_type::boolean;
        (void)m->ptr->set_attribute(changed_ind[i], bvar);
    }
    if(!c->analyze())
    {
        set_err("analyze_circuit: " + c->last_error);
        return 1;
    }
    return circuit_sample(circuit_ptr, vec_pos, chunk_pos, comp_size, voltage, voltage_ord, current, current_ord, digital, digital_ord);
}

}  // extern "C"
