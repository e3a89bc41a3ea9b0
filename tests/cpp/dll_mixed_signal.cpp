// Mixed-signal through the loader: a resistive divider feeds a comparator (code 19) whose digital output is gated with a
// digital INPUT (200) by AND / NAND / XOR (204, 208, 206) into OUTPUT probes (201); circuit_analyze + circuit_digital_clk,
// 4-state read-back with circuit_sample_digital_state_u8 and INPUT toggled with circuit_set_model_digital.
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include <phy_engine_dll_api.h>

int main()
{
    // 0 gnd | 1 VDC 5 V | 2 R 1k | 3 R 1k | 4 VDC 2 V (reference) | 5 comparator | 6 INPUT (H) | 7 AND | 8 NAND | 9 XOR | 10..12 OUTPUT
    int elements[] = {0, 4, 1, 1, 4, 19, 200, 204, 208, 206, 201, 201, 201};
    double properties[] = {5.0, 1000.0, 1000.0, 2.0, /* comparator Ll, Hl */ 0.0, 5.0, /* INPUT state */ 1.0};
    int wires[] = {
        1, 0, 2, 0,  // 5 V - R1 A
        1, 1, 0, 0,
        2, 1, 3, 0,  // divider tap (2.5 V)
        3, 1, 0, 0,
        4, 0, 5, 1,  // 2 V - comparator B (inverting)
        4, 1, 0, 0,
        5, 0, 2, 1,  // comparator A (non-inverting) - tap
        5, 2, 7, 0,  // comparator o - AND ia
        5, 2, 8, 0,  //              - NAND ia
        5, 2, 9, 0,  //              - XOR ia
        6, 0, 7, 1,  // INPUT o - AND ib
        6, 0, 8, 1,
        6, 0, 9, 1,
        7, 2, 10, 0,  // gate outputs - probes
        8, 2, 11, 0,
        9, 2, 12, 0,
    };
    std::size_t *vec_pos{}, *chunk_pos{}, comp_size{};
    void* c = create_circuit(elements, sizeof(elements) / sizeof(int), wires, sizeof(wires) / sizeof(int), properties, &vec_pos, &chunk_pos, &comp_size);
    if(!c || comp_size != 12)
    {
        std::fprintf(stderr, "dll_mixed_signal: create_circuit: %s\n", phy_engine_last_error());
        return 1;
    }
    if(circuit_set_analyze_type(c, 1 /* DC */) != 0) return 1;
    double voltage[64]{}, current[64]{};
    std::uint8_t digital[64]{};
    std::size_t voltage_ord[13]{}, current_ord[13]{}, digital_ord[13]{};
    auto settle = [&]() -> int
    {
        if(circuit_analyze(c) != 0) return 1;
        for(int k = 0; k < 3; ++k)
            if(circuit_digital_clk(c) != 0) return 1;
        return circuit_sample_digital_state_u8(c, vec_pos, chunk_pos, comp_size, voltage, voltage_ord, current, current_ord, digital, digital_ord);
    };
    if(settle() != 0)
    {
        std::fprintf(stderr, "dll_mixed_signal: %s\n", phy_engine_last_error());
        return 2;
    }
    // components: 0 VDC, 1 R, 2 R, 3 VDC, 4 comparator (A, B, o), 5 INPUT, 6 AND, 7 NAND, 8 XOR, 9..11 OUTPUT
    auto probe = [&](int comp) { return digital[digital_ord[comp]]; };
    // tap 2.5 V >= 2 V -> comparator H; INPUT H: AND = H(1), NAND = L(0), XOR = L(0)
    if(probe(9) != 1 || probe(10) != 0 || probe(11) != 0)
    {
        std::fprintf(stderr, "dll_mixed_signal: first state %u %u %u\n", probe(9), probe(10), probe(11));
        return 3;
    }
    // INPUT -> L: AND = L, NAND = H, XOR = H
    if(circuit_set_model_digital(c, vec_pos[5], chunk_pos[5], 0, 0) != 0) return 4;
    if(settle() != 0) return 5;
    if(probe(9) != 0 || probe(10) != 1 || probe(11) != 1) return 6;
    // reference raised above the tap: comparator L -> AND = L, NAND = H, XOR = L (L ^ L)
    if(circuit_set_model_double_by_name(c, vec_pos[3], chunk_pos[3], "V", 1, 3.0) != 0) return 7;
    if(settle() != 0) return 8;
    if(probe(9) != 0 || probe(10) != 1 || probe(11) != 0) return 9;
    // INPUT -> X: AND with L stays L (L dominates), NAND = H, XOR = X (2)
    if(circuit_set_model_digital(c, vec_pos[5], chunk_pos[5], 0, 2) != 0) return 10;
    if(settle() != 0) return 11;
    if(probe(9) != 0 || probe(10) != 1 || probe(11) != 2)
    {
        std::fprintf(stderr, "dll_mixed_signal: X state %u %u %u\n", probe(9), probe(10), probe(11));
        return 12;
    }
    destroy_circuit(c, vec_pos, chunk_pos);
    return 0;
}
