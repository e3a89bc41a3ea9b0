/* phy_engine_dll_api.h -- the FFI netlist loader + control subset of Phy-Engine's C ABI, exported by libpe_hip.so with
 * the reference's names, argument meaning and error behaviour (include/phy_engine/dll_api.h:43-47,51-135,143-250;
 * implementation restated from the behaviour of src/dll_main.cpp:1522-1700, 2254-2359, 2492-2602, 2861-2934).
 *
 * Subset (SURVEY.md 8b "FFI loader"): element codes 1 R, 2 C, 3 L, 4 VDC, 5 VAC{Vp, f[Hz], phase[deg]}, 6 IDC,
 * 13 PN_junction{Is,N,Isr,Nr,Temp,Ibv,Bv,Bv_set,Area}, 54 full_bridge_rectifier, 0 = ground placeholder, and (SURVEY.md 8f
 * rank 1) 7 IAC{Ip, f[Hz], phase[deg]}, 8 VCCS{G}, 9 VCVS{Mu}, 10 CCCS{alpha}, 11 CCVS{r}, 12 switch{cut_through},
 * 14 transformer{n}, 15 coupled inductors{L1,L2,k}, 16 center-tap transformer{n_total}, 17 op-amp{mu}, 18 relay{Von,Voff}, 20 sawtooth{Vh,Vl,f,phase}, 21 square{Vh,Vl,f,duty,phase},
 * 22 pulse{Vh,Vl,f,duty,phase,tr,tf}, 23 triangle{Vh,Vl,f,phase}, 50/51 BJT NPN/PNP{Is,N,BetaF,Temp,Area},
 * 52/53 level-1 N/PMOSFET{Kp,lambda,Vth}.  Every analysis runs on the MI355X through include/pe_hip.h.  Other element codes
 * Mixed-signal: 19 comparator{Ll,Hl}, 200 INPUT{state}, 201 OUTPUT, 202 OR, 203 YES, 204 AND, 205 NOT, 206 XOR, 207 XNOR, 208 NAND,
 * 209 NOR, 210 TRI, 211 IMP, 212 NIMP, 220 HALF_ADDER, 221 FULL_ADDER, 222 HALF_SUB, 223 FULL_SUB, 224 MUL2, 225 DFF, 226 TFF,
 * 227 T_BAR_FF, 228 JKFF, 229 COUNTER4{init_value}, 230 RANDOM_GENERATOR4{init_state}, 231 EIGHT_BIT_INPUT{value},
 * 232 EIGHT_BIT_DISPLAY, 233 SCHMITT_TRIGGER{Vth_low,Vth_high,inverted,Ll,Hl} (event logic on the host, circuit_digital_clk).
 * Other element codes (BSIM3 = 24, Verilog 300-301 through create_circuit_ex) are rejected.
 */
#ifndef PHY_ENGINE_DLL_API_SUBSET_H
#define PHY_ENGINE_DLL_API_SUBSET_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

char const* phy_engine_last_error(void); /* thread-local, valid until the next API call on this thread (dll_api.h:43-47) */
void phy_engine_clear_error(void);

/* dll_api.h:143-150.  Returns NULL on failure; *vec_pos / *chunk_pos are malloc'd and released by destroy_circuit. */
void* create_circuit(int* elements, size_t ele_size, int* wires, size_t wires_size, double* properties, size_t** vec_pos, size_t** chunk_pos,
                     size_t* comp_size);
/* dll_api.h:156-168: create_circuit plus a string table for Verilog sources.  The string table is accepted and ignored; a netlist
 * that contains element code 300 / 301 (Verilog module / synthesised netlist: out of scope) is refused with a
 * message, every other netlist is built exactly as create_circuit builds it. */
void* create_circuit_ex(int* elements, size_t ele_size, int* wires, size_t wires_size, double* properties, char const* const* texts,
                        size_t const* text_sizes, size_t text_count, size_t const* element_src_index, size_t const* element_top_index,
                        size_t** vec_pos, size_t** chunk_pos, size_t* comp_size);
void destroy_circuit(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos);

int circuit_set_analyze_type(void* circuit_ptr, uint32_t analyze_type_value); /* 0 OP, 1 DC, 2 AC, 3 ACOP, 4 TR, 5 TROP */
int circuit_set_tr(void* circuit_ptr, double t_step, double t_stop);
int circuit_set_ac_omega(void* circuit_ptr, double omega); /* single-point AC at omega [rad/s]; samples report the real parts */
int circuit_set_temperature(void* circuit_ptr, double temp_c);
int circuit_set_tnom(void* circuit_ptr, double tnom_c);
int circuit_set_model_double_by_name(void* circuit_ptr, size_t vec_pos, size_t chunk_pos, char const* name, size_t name_size, double value);
int circuit_analyze(void* circuit_ptr); /* 0 ok, 1 analysis failed */
int circuit_digital_clk(void* circuit_ptr);

int circuit_sample_layout(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, size_t* voltage_ord, size_t* current_ord,
                          size_t* digital_ord);
int circuit_sample(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord, double* current,
                   size_t* current_ord, bool* digital, size_t* digital_ord);
int circuit_sample_u8(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord, double* current,
                      size_t* current_ord, uint8_t* digital, size_t* digital_ord);
/* dll_api.h:222-234: digital pins report their 4-state value (0 L, 1 H, 2 X, 3 Z); set one digital attribute of a component */
int circuit_sample_digital_state_u8(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, double* voltage, size_t* voltage_ord,
                                    double* current, size_t* current_ord, uint8_t* digital, size_t* digital_ord);
int circuit_set_model_digital(void* circuit_ptr, size_t vec_pos, size_t chunk_pos, size_t attribute_index, uint8_t state);
void phy_engine_string_free(char* s); /* dll_api.h:47 (nothing in this subset returns an owned string; kept for loaders that resolve it) */
int analyze_circuit(void* circuit_ptr, size_t* vec_pos, size_t* chunk_pos, size_t comp_size, int* changed_ele, size_t* changed_ind, double* changed_prop,
                    size_t prop_size, double* voltage, size_t* voltage_ord, double* current, size_t* current_ord, bool* digital, size_t* digital_ord);

#ifdef __cplusplus
}
#endif
#endif
