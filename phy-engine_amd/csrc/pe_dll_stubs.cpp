// pe_dll_stubs.cpp -- the part of the reference's C ABI that lies OUTSIDE this engine's scope, exported so that existing
// clients can bind the whole symbol table of `include/phy_engine/dll_api.h` (the ctypes client python/phy_engine/_ffi.py sets
// argtypes on all 90 symbols eagerly and raises AttributeError on the first missing one) -- and refused loudly when called:
//   * Verilog runtime (dll_api.h:270-313) and the PhysicsLab bridge (dll_api.h:315-409): not part of the transient hot path
//     (out of scope); every entry point sets phy_engine_last_error() and returns the ABI's failure value
//     (null handle / non-zero status / 0 size);
//   * the eight Verilog-synthesis option knobs (dll_api.h:253-268) are plain process-wide settings in the reference
//     (src/dll_main.cpp:58-61 defaults): kept as stored values with those defaults, because clients set them unconditionally
//     at start-up; they affect nothing here (create_circuit_ex refuses Verilog elements).
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <string>

void pe_dll_set_error(std::string const& s);

namespace
{
    std::atomic<std::uint8_t> g_opt_level{0};
    std::atomic<bool> g_assume_binary{false}, g_allow_inout{true}, g_allow_multi_driver{true}, g_opt_wires{true}, g_opt_mul2{true}, g_opt_adders{true};
    std::atomic<std::size_t> g_loop_unroll{64};

    template <class T>
    T refuse(char const* fn, T failure)
    {
        pe_dll_set_error(std::string(fn) + ": not available in the MI355X transient engine (Verilog runtime and PhysicsLab bridge are outside its scope)");
        return failure;
    }
}  // namespace

#define PE_REFUSE(value) return refuse(__func__, value)

extern "C" {

void verilog_synth_set_opt_level(std::uint8_t level) { g_opt_level = level; }
std::uint8_t verilog_synth_get_opt_level(void) { return g_opt_level; }
void verilog_synth_set_assume_binary_inputs(bool v) { g_assume_binary = v; }
bool verilog_synth_get_assume_binary_inputs(void) { return g_assume_binary; }
void verilog_synth_set_allow_inout(bool v) { g_allow_inout = v; }
bool verilog_synth_get_allow_inout(void) { return g_allow_inout; }
void verilog_synth_set_allow_multi_driver(bool v) { g_allow_multi_driver = v; }
bool verilog_synth_get_allow_multi_driver(void) { return g_allow_multi_driver; }
void verilog_synth_set_optimize_wires(bool v) { g_opt_wires = v; }
bool verilog_synth_get_optimize_wires(void) { return g_opt_wires; }
void verilog_synth_set_optimize_mul2(bool v) { g_opt_mul2 = v; }
bool verilog_synth_get_optimize_mul2(void) { return g_opt_mul2; }
void verilog_synth_set_optimize_adders(bool v) { g_opt_adders = v; }
bool verilog_synth_get_optimize_adders(void) { return g_opt_adders; }
void verilog_synth_set_loop_unroll_limit(std::size_t n) { g_loop_unroll = n; }
std::size_t verilog_synth_get_loop_unroll_limit(void) { return g_loop_unroll; }

// ---- Verilog runtime (dll_api.h:270-313): handle creation fails, so no other entry point can receive a live handle
void* verilog_runtime_create(char const*, std::size_t, char const*, std::size_t, char const* const*, std::size_t const*, std::size_t) { PE_REFUSE(static_cast<void*>(nullptr)); }
void verilog_runtime_destroy(void*) {}
std::uint64_t verilog_runtime_get_tick(void*) { PE_REFUSE(std::uint64_t{0}); }
int verilog_runtime_reset(void*) { PE_REFUSE(1); }
int verilog_runtime_step(void*, std::uint64_t, std::uint8_t) { PE_REFUSE(1); }
int verilog_runtime_tick(void*) { PE_REFUSE(1); }
std::size_t verilog_runtime_module_count(void*) { PE_REFUSE(std::size_t{0}); }
std::size_t verilog_runtime_port_count(void*) { PE_REFUSE(std::size_t{0}); }
std::size_t verilog_runtime_signal_count(void*) { PE_REFUSE(std::size_t{0}); }
std::size_t verilog_runtime_preprocessed_size(void*) { PE_REFUSE(std::size_t{0}); }
int verilog_runtime_copy_preprocessed(void*, char*, std::size_t) { PE_REFUSE(1); }
std::size_t verilog_runtime_top_module_name_size(void*) { PE_REFUSE(std::size_t{0}); }
int verilog_runtime_copy_top_module_name(void*, char*, std::size_t) { PE_REFUSE(1); }
std::size_t verilog_runtime_module_name_size(void*, std::size_t) { PE_REFUSE(std::size_t{0}); }
int verilog_runtime_copy_module_name(void*, std::size_t, char*, std::size_t) { PE_REFUSE(1); }
std::size_t verilog_runtime_port_name_size(void*, std::size_t) { PE_REFUSE(std::size_t{0}); }
int verilog_runtime_copy_port_name(void*, std::size_t, char*, std::size_t) { PE_REFUSE(1); }
std::uint8_t verilog_runtime_port_dir(void*, std::size_t) { PE_REFUSE(std::uint8_t{0}); }
std::uint8_t verilog_runtime_get_port_value(void*, std::size_t) { PE_REFUSE(std::uint8_t{2}); }  // 2 = X
int verilog_runtime_set_port_value(void*, std::size_t, std::uint8_t) { PE_REFUSE(1); }
std::size_t verilog_runtime_signal_name_size(void*, std::size_t) { PE_REFUSE(std::size_t{0}); }
int verilog_runtime_copy_signal_name(void*, std::size_t, char*, std::size_t) { PE_REFUSE(1); }
std::uint8_t verilog_runtime_get_signal_value(void*, std::size_t) { PE_REFUSE(std::uint8_t{2}); }
int verilog_runtime_set_signal_value(void*, std::size_t, std::uint8_t) { PE_REFUSE(1); }

// ---- PhysicsLab experiment handle (dll_api.h:315-377)
void* pl_experiment_create(int) { PE_REFUSE(static_cast<void*>(nullptr)); }
void* pl_experiment_load_from_string(char const*, std::size_t) { PE_REFUSE(static_cast<void*>(nullptr)); }
void* pl_experiment_load_from_file(char const*, std::size_t) { PE_REFUSE(static_cast<void*>(nullptr)); }
void pl_experiment_destroy(void*) {}
char* pl_experiment_dump(void*, int) { PE_REFUSE(static_cast<char*>(nullptr)); }
int pl_experiment_save(void*, char const*, std::size_t, int) { PE_REFUSE(1); }
char* pl_experiment_add_circuit_element(void*, char const*, std::size_t, double, double, double, std::uint8_t, std::uint8_t, std::uint8_t) { PE_REFUSE(static_cast<char*>(nullptr)); }
int pl_experiment_connect(void*, char const*, std::size_t, int, char const*, std::size_t, int, int) { PE_REFUSE(1); }
int pl_experiment_clear_wires(void*) { PE_REFUSE(1); }
int pl_experiment_set_xyz_precision(void*, int) { PE_REFUSE(1); }
int pl_experiment_set_element_xyz(void*, std::uint8_t, double, double, double) { PE_REFUSE(1); }
int pl_experiment_set_camera(void*, double, double, double, double, double, double) { PE_REFUSE(1); }
int pl_experiment_set_element_property_number(void*, char const*, std::size_t, char const*, std::size_t, double) { PE_REFUSE(1); }
int pl_experiment_set_element_label(void*, char const*, std::size_t, char const*, std::size_t) { PE_REFUSE(1); }
int pl_experiment_set_element_position(void*, char const*, std::size_t, double, double, double, std::uint8_t) { PE_REFUSE(1); }
int pl_experiment_merge(void*, void*, double, double, double) { PE_REFUSE(1); }

// ---- PhysicsLab -> PE simulation handle (dll_api.h:379-409)
void* pl_pe_circuit_build(void*) { PE_REFUSE(static_cast<void*>(nullptr)); }
void pl_pe_circuit_destroy(void*) {}
std::size_t pl_pe_circuit_comp_size(void*) { PE_REFUSE(std::size_t{0}); }
int pl_pe_circuit_set_analyze_type(void*, std::uint32_t) { PE_REFUSE(1); }
int pl_pe_circuit_set_tr(void*, double, double) { PE_REFUSE(1); }
int pl_pe_circuit_set_ac_omega(void*, double) { PE_REFUSE(1); }
int pl_pe_circuit_analyze(void*) { PE_REFUSE(1); }
int pl_pe_circuit_digital_clk(void*) { PE_REFUSE(1); }
int pl_pe_circuit_sync_inputs_from_pl(void*, void*) { PE_REFUSE(1); }
int pl_pe_circuit_write_back_to_pl(void*, void*) { PE_REFUSE(1); }
int pl_pe_circuit_write_back_to_pl_ex(void*, void*, double, double, double, double) { PE_REFUSE(1); }
int pl_pe_circuit_sample_layout(void*, std::size_t*, std::size_t*, std::size_t*) { PE_REFUSE(1); }
int pl_pe_circuit_sample_u8(void*, double*, std::size_t*, double*, std::size_t*, std::uint8_t*, std::size_t*) { PE_REFUSE(1); }
int pl_pe_circuit_sample_digital_state_u8(void*, double*, std::size_t*, double*, std::size_t*, std::uint8_t*, std::size_t*) { PE_REFUSE(1); }

// ---- PE -> PhysicsLab export (dll_api.h:411-419)
void* pe_to_pl_convert(void*, double, double, double, std::uint8_t, std::uint8_t, std::uint8_t, std::uint8_t, std::uint8_t, std::uint8_t, std::uint8_t) { PE_REFUSE(static_cast<void*>(nullptr)); }

// ---- PhysicsLab auto-layout (dll_api.h:421-440)
int pl_experiment_auto_layout(void*, double, double, double, double, double, double, double, int, int, double, double, double, double, std::size_t*, std::size_t*,
                              std::size_t*, std::size_t*, std::size_t*)
{
    PE_REFUSE(1);
}

}  // extern "C"
