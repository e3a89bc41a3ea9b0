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

/* ---- Outside this engine's scope, exported so that clients binding the reference's WHOLE symbol table load (the ctypes client
 * python/phy_engine/_ffi.py sets argtypes on all 90 symbols eagerly); implemented in pe_dll_stubs.cpp:
 *   - Verilog-synthesis option knobs, dll_api.h:253-268: stored process-wide values with the reference's defaults
 *     (src/dll_main.cpp:58-61); they influence nothing here (create_circuit_ex refuses Verilog elements);
 *   - Verilog runtime (dll_api.h:270-313), PhysicsLab experiment / circuit handles (dll_api.h:315-409), pe_to_pl_convert and
 *     pl_experiment_auto_layout (dll_api.h:411-440): every call REFUSES -- sets phy_engine_last_error() and returns the failure value of its type
 *     (null handle, non-zero status, size 0, digital state X). */
void verilog_synth_set_opt_level(uint8_t level);
uint8_t verilog_synth_get_opt_level(void);
void verilog_synth_set_assume_binary_inputs(bool value);
bool verilog_synth_get_assume_binary_inputs(void);
void verilog_synth_set_allow_inout(bool value);
bool verilog_synth_get_allow_inout(void);
void verilog_synth_set_allow_multi_driver(bool value);
bool verilog_synth_get_allow_multi_driver(void);
void verilog_synth_set_optimize_wires(bool value);
bool verilog_synth_get_optimize_wires(void);
void verilog_synth_set_optimize_mul2(bool value);
bool verilog_synth_get_optimize_mul2(void);
void verilog_synth_set_optimize_adders(bool value);
bool verilog_synth_get_optimize_adders(void);
void verilog_synth_set_loop_unroll_limit(size_t n);
size_t verilog_synth_get_loop_unroll_limit(void);
void* verilog_runtime_create(char const* src, size_t src_size, char const* top, size_t top_size, char const* const* include_dirs,
                             size_t const* include_dir_sizes, size_t include_dir_count);
void verilog_runtime_destroy(void* runtime_ptr);
uint64_t verilog_runtime_get_tick(void* runtime_ptr);
int verilog_runtime_reset(void* runtime_ptr);
int verilog_runtime_step(void* runtime_ptr, uint64_t tick, uint8_t process_sequential);
int verilog_runtime_tick(void* runtime_ptr);
size_t verilog_runtime_module_count(void* runtime_ptr);
size_t verilog_runtime_port_count(void* runtime_ptr);
size_t verilog_runtime_signal_count(void* runtime_ptr);
size_t verilog_runtime_preprocessed_size(void* runtime_ptr);
int verilog_runtime_copy_preprocessed(void* runtime_ptr, char* out, size_t out_size);
size_t verilog_runtime_top_module_name_size(void* runtime_ptr);
int verilog_runtime_copy_top_module_name(void* runtime_ptr, char* out, size_t out_size);
size_t verilog_runtime_module_name_size(void* runtime_ptr, size_t module_index);
int verilog_runtime_copy_module_name(void* runtime_ptr, size_t module_index, char* out, size_t out_size);
size_t verilog_runtime_port_name_size(void* runtime_ptr, size_t port_index);
int verilog_runtime_copy_port_name(void* runtime_ptr, size_t port_index, char* out, size_t out_size);
uint8_t verilog_runtime_port_dir(void* runtime_ptr, size_t port_index);
uint8_t verilog_runtime_get_port_value(void* runtime_ptr, size_t port_index);
int verilog_runtime_set_port_value(void* runtime_ptr, size_t port_index, uint8_t state);
size_t verilog_runtime_signal_name_size(void* runtime_ptr, size_t signal_index);
int verilog_runtime_copy_signal_name(void* runtime_ptr, size_t signal_index, char* out, size_t out_size);
uint8_t verilog_runtime_get_signal_value(void* runtime_ptr, size_t signal_index);
int verilog_runtime_set_signal_value(void* runtime_ptr, size_t signal_index, uint8_t state);
void* pl_experiment_create(int type_value);
void* pl_experiment_load_from_string(char const* sav_json, size_t sav_json_size);
void* pl_experiment_load_from_file(char const* path, size_t path_size);
void pl_experiment_destroy(void* experiment_ptr);
char* pl_experiment_dump(void* experiment_ptr, int indent);
int pl_experiment_save(void* experiment_ptr, char const* path, size_t path_size, int indent);
char* pl_experiment_add_circuit_element(void* experiment_ptr, char const* model_id, size_t model_id_size, double x, double y, double z,
                                        uint8_t element_xyz_coords, uint8_t is_big_element, uint8_t participate_in_layout);
int pl_experiment_connect(void* experiment_ptr, char const* src_id, size_t src_id_size, int src_pin, char const* dst_id, size_t dst_id_size, int dst_pin,
                          int color_value);
int pl_experiment_clear_wires(void* experiment_ptr);
int pl_experiment_set_xyz_precision(void* experiment_ptr, int decimals);
int pl_experiment_set_element_xyz(void* experiment_ptr, uint8_t enabled, double origin_x, double origin_y, double origin_z);
int pl_experiment_set_camera(void* experiment_ptr, double vision_center_x, double vision_center_y, double vision_center_z, double target_rotation_x,
                             double target_rotation_y, double target_rotation_z);
int pl_experiment_set_element_property_number(void* experiment_ptr, char const* element_id, size_t element_id_size, char const* key, size_t key_size,
                                              double value);
int pl_experiment_set_element_label(void* experiment_ptr, char const* element_id, size_t element_id_size, char const* label, size_t label_size);
int pl_experiment_set_element_position(void* experiment_ptr, char const* element_id, size_t element_id_size, double x, double y, double z,
                                       uint8_t element_xyz_coords);
int pl_experiment_merge(void* dst_experiment_ptr, void* src_experiment_ptr, double offset_x, double offset_y, double offset_z);
void* pl_pe_circuit_build(void* experiment_ptr);
void pl_pe_circuit_destroy(void* pe_circuit_ptr);
size_t pl_pe_circuit_comp_size(void* pe_circuit_ptr);
int pl_pe_circuit_set_analyze_type(void* pe_circuit_ptr, uint32_t analyze_type_value);
int pl_pe_circuit_set_tr(void* pe_circuit_ptr, double t_step, double t_stop);
int pl_pe_circuit_set_ac_omega(void* pe_circuit_ptr, double omega);
int pl_pe_circuit_analyze(void* pe_circuit_ptr);
int pl_pe_circuit_digital_clk(void* pe_circuit_ptr);
int pl_pe_circuit_sync_inputs_from_pl(void* pe_circuit_ptr, void* experiment_ptr);
int pl_pe_circuit_write_back_to_pl(void* pe_circuit_ptr, void* experiment_ptr);
int pl_pe_circuit_write_back_to_pl_ex(void* pe_circuit_ptr, void* experiment_ptr, double logic_output_low, double logic_output_high, double logic_output_x,
                                      double logic_output_z);
int pl_pe_circuit_sample_layout(void* pe_circuit_ptr, size_t* voltage_ord, size_t* current_ord, size_t* digital_ord);
int pl_pe_circuit_sample_u8(void* pe_circuit_ptr, double* voltage, size_t* voltage_ord, double* current, size_t* current_ord, uint8_t* digital,
                            size_t* digital_ord);
int pl_pe_circuit_sample_digital_state_u8(void* pe_circuit_ptr, double* voltage, size_t* voltage_ord, double* current, size_t* current_ord,
                                          uint8_t* digital, size_t* digital_ord);
void* pe_to_pl_convert(void* circuit_ptr, double fixed_x, double fixed_y, double fixed_z, uint8_t element_xyz_coords, uint8_t keep_pl_macros,
                       uint8_t include_linear, uint8_t include_ground, uint8_t generate_wires, uint8_t keep_unknown_as_placeholders,
                       uint8_t drop_dangling_logic_inputs);
int pl_experiment_auto_layout(void* experiment_ptr, double corner0_x, double corner0_y, double corner0_z, double corner1_x, double corner1_y, double corner1_z,
                              double z_fixed, int backend_value, int mode_value, double step_x, double step_y, double margin_x, double margin_y,
                              size_t* out_grid_w, size_t* out_grid_h, size_t* out_fixed_obstacles, size_t* out_placed, size_t* out_skipped);

#ifdef __cplusplus
}
#endif
#endif
