// pe_engine_internal.hpp -- what the translation units of the engine share: the engine object behind the opaque pe_hip_engine handle of
// include/pe_hip.h, its device-memory pools, the error convention, and the helpers that cross file boundaries.  Not installed, not part
// of the C ABI.
//   pe_engine.cpp             handle life cycle, options, circuit load, getters / setters, the solve_csr_real seam
//   pe_engine_policy.cpp      launch geometry by batch size, symbolic analysis + its upload (the PHY_ENGINE_HIP_* knobs live here)
//   pe_engine_newton.cpp      host-driven Newton / transient loops of the split schedule, residual safety net, pe_hip_analyze_tr / _dc
//   pe_engine_checkpoint.cpp  pe_hip_checkpoint_*
//   pe_engine_ac.cpp          pe_hip_analyze_ac / pe_hip_get_solution_ac
//   pe_engine_seam.cpp        pe_hip_solve_csr_complex (the complex twin of the solver seam)
#pragma once
// (the helpers below are shared between the engine's translation units only: not exported from libpe_hip.so)
#define PE_ENG_HIDDEN __attribute__((visibility("hidden")))
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pe_hip.h"
#include "pe_ac.hpp"
#include "pe_circuit.hpp"
#include "pe_device.hpp"
#include "pe_kernels.hpp"
#include "pe_symbolic.hpp"

#include <map>
#include <thread>

namespace pe_eng  // (Pool is a member type of the engine object: default visibility, header-only)
{
    using clk = std::chrono::steady_clock;
    inline double ms_since(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

    struct Pool
    {
        std::vector<void*> ptrs;
        size_t bytes{};
        ~Pool() { release(); }
        void release()
        {
            for(void* p: ptrs) (void)hipFree(p);
            ptrs.clear();
            bytes = 0;
        }
        template <class T>
        hipError_t alloc(T*& out, size_t n, bool zero = true)
        {
            out = nullptr;
            size_t const b = std::max<size_t>(n, 1) * sizeof(T);
            void* p{};
            hipError_t e = hipMalloc(&p, b);
            if(e != hipSuccess) return e;
            ptrs.push_back(p);
            bytes += b;
            if(zero)
            {
                e = hipMemset(p, 0, b);
                if(e != hipSuccess) return e;
            }
            out = static_cast<T*>(p);
            return hipSuccess;
        }
        template <class T>
        hipError_t upload(T const*& out, std::vector<T> const& v)
        {
            T* p{};
            hipError_t e = alloc(p, v.size(), false);
            if(e != hipSuccess) return e;
            if(!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
            out = p;
            return e;
        }
    };
}  // namespace pe_eng
using pe_eng::Pool;

struct pe_hip_engine
{
    int device{};
    hipStream_t stream{};
    hipEvent_t ev0{}, ev1{};
    hipEvent_t evk0{}, evk1{};  // around the dominant launch of one split-schedule iteration
    double dominant_ms{};       // accumulated over the current analyze call
    int dominant_launches{};
    std::string err;
    pe_hip_options opt{};
    int lds_limit{65536};
    std::map<std::string, int> knobs;  // pe_hip_set_knob: this engine's overrides of the PHY_ENGINE_HIP_* tuning knobs (name without the prefix)

    // resident circuit
    bool loaded{};
    pe::HostCircuit hc;
    std::vector<int> drv_node;
    std::vector<double> drv_volt;
    pe::OverlaySpec overlay;      // host-stamp overlay (pe_hip_set_overlay): part of the pattern of the next load
    pe_hip_overlay_fn overlay_fn{};
    void* overlay_user{};
    std::vector<double> ov_x, ov_a, ov_b;  // staging of the callback
    bool singular_rematched{};    // the one re-match after a singular pivot has been spent for this resident circuit
    bool careful{};               // residual safety net tripped on the resident kernel: stay on the host-driven (refining) schedule
    long long n_refined{}, n_rematched{};  // solves repaired by refinement / symbolic re-analyses on an instance's own values (diagnostics)
    // host-driven Newton loop (split schedule): pinned staging for the per-iteration `active` upload / `flags` read-back, and what
    // the device's `active` array currently holds (an unchanged mask is not uploaded again)
    int* pin_active{};
    int* pin_flags{};
    size_t pin_cap{};
    // results published by the last launch of an iteration straight into pinned host memory (k_m2_publish): [sequence number][flags][norms]
    void* pub_host{};
    void* pub_dev{};  // the same memory as the device sees it
    size_t pub_cap{};
    unsigned long long pub_seq{};
    std::vector<int> active_dev;
    // captured launch sequences of the split schedule's Newton iteration (small sweeps; pe_kernels.hip launch_m2_iteration_graph)
    pe::M2GraphCache* graphs{};
    bool graph_mode{};            // the quad list behind the device's `active` mask is laid out for a captured sequence (full grid)
    double* stats_scratch{};      // pe_hip_sweep_statistics: partial sums + result (device, owned by circ_pool)
    size_t stats_doubles{};
    double* stats_pinned{};       // pinned host landing buffer of its result (owned by the engine)
    size_t stats_pinned_bytes{};
    Pool circ_pool;  // topology, params, state
    Pool sym_pool;   // symbolic arrays + factor storage
    pe::Symbolic sym;
    int sym_class{-1};  // 0: static (OP/DC/TROP) pattern weights, 1: TR
    double sym_dt{};    // time step whose companion values the TR analysis was matched on
    pe::DevView V{};
    bool fact_valid{};
    double fact_dt{};
    // split schedule: instances whose matrix (aval) holds a FULL transient stamp at step size a_static_dt -- their next steps at that dt stamp
    // the x-dependent slots only (pe_engine_newton.cpp run_m2_tr); cleared together with fact_valid wherever a static value may change
    std::vector<char> a_static;
    double a_static_dt{-1.0};
    double analyze_ms{};

    // small-signal AC: a second engine holding the real-equivalent 2N system (pe_ac.hpp), built on first use
    struct Ac
    {
        pe_hip_engine* eng{};
        pe::AcCircuit circ;
        bool built{};
        double sym_omega{-1.0};  // frequency whose values the pivot matching of the current symbolic analysis saw
        std::vector<int> b_ptr0, b_src0;  // right-hand-side lists of the AC system (the device copy reads one slot per row)
        int rhs0{};                       // first of the 2N right-hand-side slots of the AC value vector
        std::vector<double> x;            // refined solution [batch][2N]
        double *d_xacc{}, *d_b0{}, *d_worst{};  // device: accumulated solution, the point's right-hand side, worst backward error (refinement)
        size_t d_len{};
    } ac;
    std::vector<double> sym_values_override;  // representative |A| values for the row matching (AC engine)

    // solve_csr_real seam (separate small state)
    struct Csr
    {
        Pool pool;
        pe::Symbolic sym;
        pe::DevView V{};
        int n{-1}, nnz{-1};
        bool have{};
    } csr;
    // ... and its complex twin (pe_engine_seam.cpp): the real-equivalent 2n system, its refinement buffers
    struct Csrz
    {
        Pool pool;
        pe::Symbolic sym;
        pe::DevView V{};
        int n{-1}, nnz{-1};
        bool have{};
        bool on_these_values{};            // the pivot order of the cached analysis was matched on the values of the current call
        std::vector<int> rp2, ci2, pos;    // real-equivalent pattern; the four positions of every complex entry
        std::vector<double> vals, rhs, x;  // host staging
        double *d_xacc{}, *d_b0{}, *d_worst{};  // (owned by pool)
    } csrz;
};

// a failed HIP call is NOT "no device" unless the runtime says so: out-of-memory at a large batch, a launch failure or a
// memcpy error are internal errors of a machine that has a GPU (callers and tests tell them apart)
static inline int hip_error_code(hipError_t e)
{
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? PE_HIP_ERR_NO_DEVICE : PE_HIP_ERR_INTERNAL;
}

#define HIPCHK(h, expr)                                                                             \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if(e__ != hipSuccess)                                                                       \
        {                                                                                           \
            (h)->err = std::string("HIP error: ") + hipGetErrorString(e__) + " at " #expr;          \
            return hip_error_code(e__);                                                             \
        }                                                                                           \
    } while(0)

// the contract of the solver seam's arrays (the reference builds them that way, circuit.h:1171-1226): 0-based CSR, row_ptr monotone from 0 to nnz,
// column indices inside [0, n) and strictly increasing inside a row.  Checked once per pattern: everything behind the seam indexes with these numbers.
static inline char const* csr_pattern_error(int n, int nnz, int const* row_ptr, int const* col_ind)
{
    if(n > 0 && (row_ptr[0] != 0 || row_ptr[n] != nnz)) return "row_ptr[0] != 0 or row_ptr[n] != nnz";
    for(int i = 0; i < n; ++i)
    {
        if(row_ptr[i + 1] < row_ptr[i]) return "row_ptr is not monotone";
        for(int e = row_ptr[i]; e < row_ptr[i + 1]; ++e)
            if(col_ind[e] < 0 || col_ind[e] >= n || (e > row_ptr[i] && col_ind[e] <= col_ind[e - 1])) return "column indices out of range or not sorted inside a row";
    }
    return nullptr;
}

namespace pe_eng PE_ENG_HIDDEN
{
    extern thread_local std::string g_create_error;

    // pe_engine.cpp
    int fail(pe_hip_engine* h, int code, std::string msg);
    double r_open_of(pe_hip_engine const* h);
    void apply_options(pe_hip_engine* h, pe::DevView& V);
    int stats_chunks(int batch);
    int finish_load(pe_hip_engine* h);
    void fill_static_dv(pe_hip_engine const* h, std::vector<double>& dv);
    int collect_stats(pe_hip_engine* h, std::vector<long long> const& steps0, std::vector<long long> const& iters0, pe_hip_run_stats* st);
    int snapshot_counters(pe_hip_engine* h, std::vector<long long>& s0, std::vector<long long>& i0);
    // pe_engine_policy.cpp
    bool split_launch(pe_hip_engine const* h);
    int env_int0(char const* name, int def);
    int knob(pe_hip_engine const* h, char const* name, int def);
    int upload_symbolic(pe_hip_engine* h, Pool& pool, pe::Symbolic& S, pe::SymbolicOptions const& so, pe::DevView& V, int batch);
    pe::SymbolicOptions symbolic_options(pe_hip_engine const* h, int batch_in, int rows, int panel_reserve = 384, int force_resident = 0);
    int analyze_fitting(pe_hip_engine* h, int batch, int geometry_rows, int n, int const* rp, int const* ci, double const* vals, pe::Symbolic& S,
                        pe::SymbolicOptions& so);
    int ensure_symbolic(pe_hip_engine* h, bool tr, double dt);
    // pe_engine_newton.cpp
    bool has_overlay(pe_hip_engine const* h);
    int overlay_call(pe_hip_engine* h, int event, int mode, double t, double dt, int b = 0);
    int overlay_call_all(pe_hip_engine* h, int event, int mode, double t, double dt, std::vector<int> const* mask);
}  // namespace pe_eng
