// The loader's element codes added with the remaining linear stampers (dll_api.h:60-97): 12 switch, 9 VCVS, 21 square generator,
// 14 transformer, through create_circuit / analyze_circuit, including a property change (switch opened) on the resident circuit.
#include <cmath>
#include <cstddef>
#include <cstdio>

#include <phy_engine_dll_api.h>

int main()
{
    // 0: ground, 1: VDC 5 V, 2: switch (closed), 3: R 1k, 4: VCVS mu 2, 5: R 2k, 6: square 4 V / 1 V (t = 0 -> Vh), 7: transformer n 2, 8: R 100
    int elements[] = {0, 4, 12, 1, 9, 1, 21, 14, 1};
    double properties[] = {5.0, 1.0, 1000.0, 2.0, 2000.0, /* square */ 4.0, 1.0, 1000.0, 0.5, 0.0, /* n */ 2.0, 100.0};
    int wires[] = {
        1, 0, 2, 0,  // VDC+ - switch A
        2, 1, 3, 0,  // switch B - R3 A
        3, 1, 0, 0,  // R3 B - gnd
        1, 1, 0, 0,  // VDC- - gnd
        4, 0, 5, 0,  // VCVS S - R5 A
        4, 1, 0, 0,  // VCVS T - gnd
        4, 2, 2, 1,  // VCVS P - switch B
        4, 3, 0, 0,  // VCVS Q - gnd
        5, 1, 0, 0,  // R5 B - gnd
        6, 0, 7, 0,  // square + - transformer P
        6, 1, 0, 0,  // square - - gnd
        7, 1, 0, 0,  // transformer Q - gnd
        7, 2, 8, 0,  // transformer S - R8 A
        7, 3, 0, 0,  // transformer T - gnd
        8, 1, 0, 0,  // R8 B - gnd
    };
    std::size_t *vec_pos{}, *chunk_pos{}, comp_size{};
    void* c = create_circuit(elements, sizeof(elements) / sizeof(int), wires, sizeof(wires) / sizeof(int), properties, &vec_pos, &chunk_pos, &comp_size);
    if(!c)
    {
        std::fprintf(stderr, "create_circuit: %s\n", phy_engine_last_error());
        return 1;
    }
    if(comp_size != 8) return 1;
    if(circuit_set_analyze_type(c, 1 /* DC */) != 0) return 1;
    double voltage[64]{}, current[64]{};
    std::size_t voltage_ord[9]{}, current_ord[9]{}, digital_ord[9]{};
    bool digital[64]{};
    if(analyze_circuit(c, vec_pos, chunk_pos, comp_size, nullptr, nullptr, nullptr, 0, voltage, voltage_ord, current, current_ord, digital, digital_ord) != 0)
    {
        std::fprintf(stderr, "analyze_circuit: %s\n", phy_engine_last_error());
        return 2;
    }
    // pins per component: VDC 2, switch 2, R 2, VCVS 4, R 2, square 2, transformer 4, R 2
    std::size_t const want_ord[9] = {0, 2, 4, 6, 10, 12, 14, 18, 20};
    for(int i = 0; i < 9; ++i)
        if(voltage_ord[i] != want_ord[i]) return 3;
    auto near = [](double a, double b, double tol) { return std::abs(a - b) <= tol; };
    if(!near(voltage[3], 5.0, 1e-12)) return 4;                 // switch B (closed)
    if(!near(voltage[6], 10.0, 1e-9)) return 5;                 // VCVS S = mu * V(P)
    if(!near(voltage[12], 4.0, 1e-12)) return 6;                // square at t = 0
    if(!near(voltage[16], 2.0, 1e-9)) return 7;                 // transformer secondary = Vp / n
    // open the switch on the resident circuit (component 1, attribute 0)
    int ce[] = {1};
    std::size_t ci[] = {0};
    double cp[] = {0.0};
    if(analyze_circuit(c, vec_pos, chunk_pos, comp_size, ce, ci, cp, 1, voltage, voltage_ord, current, current_ord, digital, digital_ord) != 0) return 8;
    double const leak = 5.0 * 1000.0 / (1e12 + 1000.0);
    if(!near(voltage[3], leak, 1e-15) || !near(voltage[6], 2.0 * leak, 1e-14)) return 9;
    destroy_circuit(c, vec_pos, chunk_pos);
    return 0;
}
