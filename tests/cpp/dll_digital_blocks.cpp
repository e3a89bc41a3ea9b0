// Digital blocks through the loader (element codes 210, 221, 225, 229-233 of dll_api.h:110-131): a 4-bit counter preset to 5 and a
// D flip-flop share a clock INPUT toggled with circuit_set_model_digital; a full adder and a tri-state buffer hang on constant
// INPUTs.  Known answers: three rising edges take the counter to 8, the flip-flop follows d on the edge, H + H + L = (s L, cout H),
// a disabled tri-state buffer leaves Z.  A 1 V / 1 k loop keeps an analog part in the netlist (every analysis runs on the GPU).
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include <phy_engine_dll_api.h>

int main()
{
    // 0 gnd | 1 VDC 1 V | 2 R 1k | 3 INPUT clk (L) | 4 COUNTER4 (init 5) | 5..8 OUTPUT q3..q0 | 9 INPUT d (H) | 10 DFF | 11 OUTPUT q
    // 12 INPUT a (H) | 13 INPUT b (H) | 14 INPUT cin (L) | 15 FULL_ADDER | 16 OUTPUT s | 17 OUTPUT cout
    // 18 INPUT en (L) | 19 TRI | 20 OUTPUT tri | 21 EIGHT_BIT_INPUT (0xC3) | 22 EIGHT_BIT_DISPLAY (b7..b4 wired) | 23 OUTPUT on b7
    // 24 SCHMITT_TRIGGER (inverted) on the 1 V analog node | 25 OUTPUT | 26 RANDOM_GENERATOR4 (state 9) on the clock | 27 OUTPUT q0
    int elements[] = {0, 4, 1, 200, 229, 201, 201, 201, 201, 200, 225, 201, 200, 200, 200, 221, 201, 201, 200, 210, 201,
                      231, 232, 201, 233, 201, 230, 201};
    double properties[] = {1.0, 1000.0, /* clk */ 0.0, /* counter init */ 5.0, /* d */ 1.0, /* a b cin */ 1.0, 1.0, 0.0, /* en */ 0.0,
                           /* 8-bit value */ 195.0, /* Schmitt: Vth_low Vth_high inverted Ll Hl */ 0.3, 0.6, 1.0, 0.0, 5.0, /* rng state */ 9.0};
    int wires[] = {
        1, 0, 2, 0, 1, 1, 0, 0, 2, 1, 0, 0,                      // analog loop
        3, 0, 4, 4,                                              // clk - counter clk (en left open = enabled)
        4, 0, 5, 0, 4, 1, 6, 0, 4, 2, 7, 0, 4, 3, 8, 0,          // q3..q0 - probes
        9, 0, 10, 0, 3, 0, 10, 1, 10, 2, 11, 0,                  // d - DFF d, clk - DFF clk, q - probe
        12, 0, 15, 0, 13, 0, 15, 1, 14, 0, 15, 2, 15, 3, 16, 0, 15, 4, 17, 0,
        12, 0, 19, 0, 18, 0, 19, 1, 19, 2, 20, 0,                // a - TRI i, en - TRI en, o - probe
        21, 0, 22, 0, 21, 1, 22, 1, 21, 2, 22, 2, 21, 3, 22, 3, 21, 0, 23, 0,  // b7..b4 of the 8-bit input - display, b7 - probe
        24, 0, 1, 0, 24, 1, 25, 0,                               // Schmitt input on the 1 V source node, output - probe
        3, 0, 26, 4, 26, 3, 27, 0,                               // clk - generator clk (reset_n open), q0 - probe
    };
    std::size_t *vec_pos{}, *chunk_pos{}, comp_size{};
    void* c = create_circuit(elements, sizeof(elements) / sizeof(int), wires, sizeof(wires) / sizeof(int), properties, &vec_pos, &chunk_pos, &comp_size);
    if(!c || comp_size != 27)
    {
        std::fprintf(stderr, "dll_digital_blocks: create_circuit: %s (%zu components)\n", phy_engine_last_error(), comp_size);
        return 1;
    }
    if(circuit_set_analyze_type(c, 1 /* DC */) != 0) return 1;
    if(circuit_analyze(c) != 0)
    {
        std::fprintf(stderr, "dll_digital_blocks: %s\n", phy_engine_last_error());
        return 2;
    }
    double voltage[128]{}, current[128]{};
    std::uint8_t digital[128]{};
    std::size_t voltage_ord[28]{}, current_ord[28]{}, digital_ord[28]{};
    auto sample = [&]() { return circuit_sample_digital_state_u8(c, vec_pos, chunk_pos, comp_size, voltage, voltage_ord, current, current_ord, digital, digital_ord); };
    auto probe = [&](int comp) { return static_cast<unsigned>(digital[digital_ord[comp]]); };
    // components: 0 VDC, 1 R, 2 clk, 3 counter, 4..7 q3..q0, 8 d, 9 DFF, 10 q, 11 a, 12 b, 13 cin, 14 FA, 15 s, 16 cout, 17 en, 18 TRI, 19 tri
    auto clock_edge = [&]() -> int
    {
        if(circuit_set_model_digital(c, vec_pos[2], chunk_pos[2], 0, 0) != 0 || circuit_digital_clk(c) != 0) return 1;
        if(circuit_set_model_digital(c, vec_pos[2], chunk_pos[2], 0, 1) != 0 || circuit_digital_clk(c) != 0) return 1;
        return 0;
    };
    if(circuit_digital_clk(c) != 0 || sample() != 0) return 3;
    if(probe(4) != 0 || probe(5) != 1 || probe(6) != 0 || probe(7) != 1)  // preset 5 = 0101
    {
        std::fprintf(stderr, "dll_digital_blocks: preset %u%u%u%u\n", probe(4), probe(5), probe(6), probe(7));
        return 4;
    }
    if(probe(15) != 0 || probe(16) != 1) return 5;  // H + H + L
    if(probe(19) != 3) return 6;                    // disabled tri-state: Z
    if(probe(10) != 0) return 7;                    // flip-flop has seen no edge yet
    // components 20 8-bit input, 21 display, 22 probe b7, 23 Schmitt, 24 probe, 25 generator, 26 probe q0
    if(probe(22) != 1) return 20;                   // 0xC3: b7 = H
    if(probe(24) != 0) return 21;                   // 1 V >= Vth_high 0.6 V -> H, inverted -> L
    if(probe(26) != 1) return 22;                   // generator state 9 = 1001: q0 = H
    for(int k = 0; k < 3; ++k)
        if(clock_edge() != 0) return 8;
    if(sample() != 0) return 9;
    if(probe(4) != 1 || probe(5) != 0 || probe(6) != 0 || probe(7) != 0)  // 5 + 3 = 8 = 1000
    {
        std::fprintf(stderr, "dll_digital_blocks: after 3 edges %u%u%u%u\n", probe(4), probe(5), probe(6), probe(7));
        return 10;
    }
    if(probe(10) != 1) return 11;  // d = H latched on the first edge
    // generator: 9 = 1001 -> (b3 ^ b2) ^ 1 = 0: 0010 -> 1: 0101 -> (0 ^ 1) ^ 1 = 0: 1010; q0 = L
    if(probe(26) != 0) return 23;
    // enable the tri-state buffer: passes a = H
    if(circuit_set_model_digital(c, vec_pos[17], chunk_pos[17], 0, 1) != 0 || circuit_digital_clk(c) != 0 || sample() != 0) return 12;
    if(probe(19) != 1) return 13;
    // d -> L: q keeps H until the next rising edge
    if(circuit_set_model_digital(c, vec_pos[8], chunk_pos[8], 0, 0) != 0 || circuit_digital_clk(c) != 0 || sample() != 0) return 14;
    if(probe(10) != 1) return 15;
    if(clock_edge() != 0 || sample() != 0) return 16;
    if(probe(10) != 0) return 17;
    destroy_circuit(c, vec_pos, chunk_pos);
    std::printf("dll_digital_blocks ok\n");
    return 0;
}
