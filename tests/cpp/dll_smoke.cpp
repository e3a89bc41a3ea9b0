// Mirrors test/0008.dll/dll_main_smoke.cpp: VDC 5 V - R 1k through the C-ABI loader (create_circuit / analyze_circuit /
// destroy_circuit): V(R.A) = 5, V(R.B) = 0, |I(VDC)| = 5 mA.
#include <cmath>
#include <cstddef>
#include <cstdio>

#include <phy_engine_dll_api.h>

int main()
{
    int elements[] = {0, 4, 1};
    int wires[] = {1, 0, 2, 0, 2, 1, 0, 0, 1, 1, 0, 0};
    double properties[] = {5.0, 1000.0};
    std::size_t* vec_pos{};
    std::size_t* chunk_pos{};
    std::size_t comp_size{};
    void* cptr = create_circuit(elements, 3, wires, 12, properties, &vec_pos, &chunk_pos, &comp_size);
    if(cptr == nullptr || vec_pos == nullptr || chunk_pos == nullptr)
    {
        std::fprintf(stderr, "create_circuit: %s\n", phy_engine_last_error());
        return 1;
    }
    if(comp_size != 2) return 1;
    if(circuit_set_analyze_type(cptr, 1 /* DC */) != 0) return 1;
    double voltage[16]{};
    std::size_t voltage_ord[3]{};
    double current[16]{};
    std::size_t current_ord[3]{};
    bool digital[16]{};
    std::size_t digital_ord[3]{};
    int const rc = analyze_circuit(cptr, vec_pos, chunk_pos, comp_size, nullptr, nullptr, nullptr, 0, voltage, voltage_ord, current, current_ord, digital, digital_ord);
    if(rc != 0)
    {
        std::fprintf(stderr, "analyze_circuit: %s\n", phy_engine_last_error());
        return 2;
    }
    // component 0 = VDC (pins +,-; 1 branch), component 1 = R (pins A,B)
    if(voltage_ord[1] != 2 || voltage_ord[2] != 4 || current_ord[1] != 1 || current_ord[2] != 1) return 3;
    if(std::abs(voltage[0] - 5.0) > 1e-12 || std::abs(voltage[1]) > 1e-12 || std::abs(voltage[2] - 5.0) > 1e-12 || std::abs(voltage[3]) > 1e-12) return 4;
    if(std::abs(std::abs(current[0]) - 5e-3) > 1e-12) return 5;
    // property update through analyze_circuit: R -> 2 kOhm => 2.5 mA
    int ce[] = {1};
    std::size_t ci[] = {0};
    double cp[] = {2000.0};
    if(analyze_circuit(cptr, vec_pos, chunk_pos, comp_size, ce, ci, cp, 1, voltage, voltage_ord, current, current_ord, digital, digital_ord) != 0) return 6;
    if(std::abs(std::abs(current[0]) - 2.5e-3) > 1e-12) return 7;
    // an unsupported element code fails with a message and NULL
    int bad[] = {0, 300};
    std::size_t *vp2{}, *cp2{}, cs2{};
    if(create_circuit(bad, 2, nullptr, 0, properties, &vp2, &cp2, &cs2) != nullptr) return 8;
    if(phy_engine_last_error()[0] == 0) return 9;
    // create_circuit_ex (dll_api.h:156-168): same netlist without Verilog elements builds; a Verilog element is refused loudly
    {
        char const* texts[] = {"module m; endmodule"};
        std::size_t sizes[] = {19}, src_index[] = {0, 0, 0}, top_index[] = {0, 0, 0};
        std::size_t *vp3{}, *cp3{}, cs3{};
        int elems[] = {0, 4, 1};
        double props[] = {5.0, 1000.0};
        int w3[] = {1, 0, 2, 0, 1, 1, 0, 0, 2, 1, 0, 0};
        void* c3 = create_circuit_ex(elems, 3, w3, 12, props, texts, sizes, 1, src_index, top_index, &vp3, &cp3, &cs3);
        if(!c3 || cs3 != 2) return 10;
        destroy_circuit(c3, vp3, cp3);
        if(create_circuit_ex(bad, 2, nullptr, 0, properties, texts, sizes, 1, src_index, top_index, &vp2, &cp2, &cs2) != nullptr) return 11;
        if(phy_engine_last_error()[0] == 0) return 12;
    }
    destroy_circuit(cptr, vec_pos, chunk_pos);
    return 0;
}
