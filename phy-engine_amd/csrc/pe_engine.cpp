// pe_engine.cpp -- C ABI (include/pe_hip.h) and host orchestration of the resident transient path.
//
// The host side does what circult::analyze()/prepare() do around the hot loop (circuit.h:179-296, 468-890):
// index, build the pattern once, decide how many steps to run; everything per time step runs in the kernels.
// There is deliberately NO CPU numeric fallback here: without a HIP device every compute entry point fails
// with PE_HIP_ERR_NO_DEVICE and a message.
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

namespace
{
    using clk = std::chrono::steady_clock;
    inline double ms_since(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

    thread_local std::string g_create_error;

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
}  // namespace

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
    std::vector<int> active_dev;
    double* stats_scratch{};      // pe_hip_sweep_statistics: partial sums + result (device, owned by circ_pool)
    size_t stats_doubles{};
    Pool circ_pool;  // topology, params, state
    Pool sym_pool;   // symbolic arrays + factor storage
    pe::Symbolic sym;
    int sym_class{-1};  // 0: static (OP/DC/TROP) pattern weights, 1: TR
    double sym_dt{};    // time step whose companion values the TR analysis was matched on
    pe::DevView V{};
    bool fact_valid{};
    double fact_dt{};
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

namespace
{
    int finish_load(pe_hip_engine* h);

    int fail(pe_hip_engine* h, int code, std::string msg)
    {
        h->err = std::move(msg);
        return code;
    }

    // multi-workgroup schedule (one launch per phase and tree level) instead of the single resident kernel
    // Large circuits always: the per-phase kernels fit their register budgets (the factor kernel spills 48 B / lane at 128 VGPRs,
    // the resident kernel 580), which outweighs ~35 launches and one host round trip per Newton iteration once an iteration
    // takes milliseconds.  Small circuits stay in the resident kernel (a time step is microseconds there).
    bool split_launch(pe_hip_engine const* h)
    {
        char const* v = std::getenv("PHY_ENGINE_HIP_SPLIT");  // knob: 1 = always split, 0 = never (resident kernel, one part)
        if(v && *v == '1') return true;
        if(v && *v == '0') return false;
        if(h->overlay_fn && (h->hc.n_ov_a || h->hc.n_ov_b)) return true;  // host-stamped models: the host drives the Newton loop
        if(h->careful) return true;  // an inaccurate solve was detected: the host-driven loop refines / re-matches
        return h->V.n_parts > 1 || h->hc.rows >= 3000;
    }

    // host-stamp overlay: one callback (+ the upload of its values for ITERATE) on the current x of instance b.  In a batch the
    // callback is told first which instance the calls that follow concern (PE_HIP_OVERLAY_INSTANCE): models with state of their own
    // (a junction's last voltage, a companion history) keep one copy per instance.
    int overlay_call(pe_hip_engine* h, int event, int mode, double t, double dt, int b = 0)
    {
        auto const& hc = h->hc;
        h->ov_x.resize(static_cast<size_t>(hc.rows));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if(hc.batch > 1 && h->overlay_fn(h->overlay_user, PE_HIP_OVERLAY_INSTANCE, b, t, dt, nullptr, nullptr, nullptr) != 0)
            return fail(h, PE_HIP_ERR_INTERNAL, "host-stamp overlay: the callback refused PE_HIP_OVERLAY_INSTANCE (it does not support batches)");
        if(hc.rows) HIPCHK(h, hipMemcpy(h->ov_x.data(), h->V.x + static_cast<size_t>(b) * hc.rows, static_cast<size_t>(hc.rows) * sizeof(double), hipMemcpyDeviceToHost));
        bool const iter = event == PE_HIP_OVERLAY_ITERATE;
        h->ov_a.assign(static_cast<size_t>(hc.n_ov_a), 0.0);
        h->ov_b.assign(static_cast<size_t>(hc.n_ov_b), 0.0);
        int const orc = h->overlay_fn(h->overlay_user, event, mode, t, dt, h->ov_x.data(), iter ? h->ov_a.data() : nullptr, iter ? h->ov_b.data() : nullptr);
        if(event == PE_HIP_OVERLAY_CONVERGED && orc == PE_HIP_OVERLAY_VETO) return PE_HIP_OVERLAY_VETO;  // (positive: not a pe_hip_status)
        if(orc != 0) return fail(h, PE_HIP_ERR_INTERNAL, "host-stamp overlay: a model hook failed");
        if(iter)
        {
            double* dv = h->V.dv + static_cast<size_t>(b) * h->V.dv_len;
            if(hc.n_ov_a) HIPCHK(h, hipMemcpy(dv + hc.dv_ova, h->ov_a.data(), static_cast<size_t>(hc.n_ov_a) * sizeof(double), hipMemcpyHostToDevice));
            if(hc.n_ov_b) HIPCHK(h, hipMemcpy(dv + hc.dv_ovb, h->ov_b.data(), static_cast<size_t>(hc.n_ov_b) * sizeof(double), hipMemcpyHostToDevice));
        }
        return PE_HIP_OK;
    }
    // the same for every instance of `mask` (null: all)
    int overlay_call_all(pe_hip_engine* h, int event, int mode, double t, double dt, std::vector<int> const* mask)
    {
        for(int b = 0; b < h->hc.batch; ++b)
            if(!mask || (*mask)[b])
                if(int const rc = overlay_call(h, event, mode, t, dt, b); rc != PE_HIP_OK) return rc;
        return PE_HIP_OK;
    }
    bool has_overlay(pe_hip_engine const* h) { return h->overlay_fn && (h->hc.n_ov_a || h->hc.n_ov_b); }

    double r_open_of(pe_hip_engine const* h) { return h->opt.r_open > 0.0 ? h->opt.r_open : 1e12; }  // circuit.h:1012

    void apply_options(pe_hip_engine* h, pe::DevView& V)
    {
        auto const& o = h->opt;
        V.v_abstol = o.v_abstol > 0.0 ? o.v_abstol : 1e-6;   // circuit.h:900-903
        V.v_reltol = o.v_reltol > 0.0 ? o.v_reltol : 1e-3;
        V.i_abstol = o.i_abstol > 0.0 ? o.i_abstol : 1e-12;
        V.i_reltol = o.i_reltol > 0.0 ? o.i_reltol : V.v_reltol;
        V.max_newton = o.max_newton > 0 ? o.max_newton : 64;
        V.keep_l21 = (o.refactor_every_solve || V.nonlinear) ? 0 : 1;  // only a linear circuit reuses its factors (separate forward pass over L21)
        V.r_open = r_open_of(h);
        V.residual_tol = o.residual_tol < 0.0 ? 0.0 : (o.residual_tol > 0.0 ? o.residual_tol : 1e-10);
    }

    // pe_hip_sweep_statistics: instance chunks of the first pass -- enough workgroups to stream x at HBM rate, few enough for a cheap second pass
    int stats_chunks(int batch) { return std::clamp(batch / 32, 1, 64); }

    // uploads symbolic arrays + allocates per-instance factor storage into `pool`, fills the symbolic part of V
    // test knob: choose the launch geometry as if the batch had this many instances
    int geometry_batch(int batch)
    {
        char const* v = std::getenv("PHY_ENGINE_HIP_GEOMETRY_BATCH");
        return v && *v ? std::max(1, std::atoi(v)) : batch;
    }

    int env_int0(char const* name, int def);

    int upload_symbolic(pe_hip_engine* h, Pool& pool, pe::Symbolic& S, pe::SymbolicOptions const& so, pe::DevView& V, int batch)
    {
        V.nfronts = S.nfronts;
        HIPCHK(h, pool.upload(V.f_col0, S.f_col0));
        HIPCHK(h, pool.upload(V.f_p, S.f_p));
        HIPCHK(h, pool.upload(V.f_u, S.f_u));
        HIPCHK(h, pool.upload(V.f_rows_ptr, S.f_rows_ptr));
        HIPCHK(h, pool.upload(V.f_rows, S.f_rows));
        HIPCHK(h, pool.upload(V.f_child_ptr, S.f_child_ptr));
        HIPCHK(h, pool.upload(V.f_child, S.f_child));
        HIPCHK(h, pool.upload(V.f_wstack, S.f_wstack));
        HIPCHK(h, pool.upload(V.f_wpar, S.f_wpar));
        HIPCHK(h, pool.upload(V.f_rel, S.f_rel));
        HIPCHK(h, pool.upload(V.f_inv_off, S.f_inv_off));
        HIPCHK(h, pool.upload(V.f_cnp, S.f_cnp));
        HIPCHK(h, pool.upload(V.f_inv, S.f_inv));
        HIPCHK(h, pool.upload(V.f_bmask, S.f_bmask));
        HIPCHK(h, pool.upload(V.f_asm_ptr, S.f_asm_ptr));
        HIPCHK(h, pool.upload(V.asm_slot, S.asm_slot));
        HIPCHK(h, pool.upload(V.asm_pos, S.asm_pos));
        HIPCHK(h, pool.upload(V.f_lptr, S.f_lptr));
        HIPCHK(h, pool.upload(V.f_uptr, S.f_uptr));
        HIPCHK(h, pool.upload(V.f_sptr, S.f_sptr));
        HIPCHK(h, pool.upload(V.row_src, S.row_src));
        HIPCHK(h, pool.upload(V.col_src, S.col_src));
        HIPCHK(h, pool.upload(V.wave_ptr, S.wave_ptr));
        HIPCHK(h, pool.upload(V.wave_list, S.wave_list));
        HIPCHK(h, pool.upload(V.coop_ptr, S.coop_ptr));
        HIPCHK(h, pool.upload(V.coop_list, S.coop_list));
        HIPCHK(h, pool.upload(V.top_ptr, S.top_ptr));
        HIPCHK(h, pool.upload(V.top_list, S.top_list));
        V.n_parts = S.n_parts;
        V.n_top_levels = static_cast<int>(S.top_ptr.size()) - 1;
        if(V.n_top_levels > 64) return fail(h, PE_HIP_ERR_INTERNAL, "assembly tree has more than 64 top levels");
        for(int l = 0; l < V.n_top_levels; ++l) V.top_cnt[l] = S.top_ptr[l + 1] - S.top_ptr[l];
        V.n_waves = so.n_waves;
        V.high_occupancy = so.shared_cu;
        V.wave_m = so.wave_m;
        V.max_m = std::max(S.max_m, 1);
        V.max_p = so.max_pivots;
        V.wave_p = so.wave_p;
        // a wavefront's slot holds its fronts whole (order <= wave_m, odd leading dimension) + the right-hand-side column -- or, with an
        // explicit wave_slot, the panels of the larger ones
        V.lds_slot = so.wave_slot > 0 ? static_cast<int>(so.wave_slot) : (pe::pe_ld(so.wave_m) + 1) * so.wave_m;
        V.lds_wave_stage = so.wave_m * so.wave_p;              // a wavefront stages the whole m x p panel of its (small) fronts
        V.lds_coop_stage = std::max(V.max_p * V.max_p, std::min(64, V.max_m) * V.max_p);
        V.lds_bstack_off = so.wave_m + V.lds_wave_stage + 64;  // t[m] + staged block + partial sums of one wavefront,
        V.lds_sslot = V.lds_bstack_off + std::max(1, S.wave_stack);  // + the backward stack (the solved vectors along one path of a wave subtree)
        V.lds_wave_stage_b = so.wave_p * so.wave_p;            // backward pass: U11 only (front_backward_lean)
        V.lds_bstack_off_b = so.wave_m + V.lds_wave_stage_b + 64;
        V.lds_sslot_b = V.lds_bstack_off_b + std::max(1, S.wave_stack);
        {
            long long need = static_cast<long long>(so.n_waves) * V.lds_slot;
            need = std::max(need, so.panel_doubles + so.panel_reserve);
            need = std::max(need, static_cast<long long>(so.n_waves) * V.lds_sslot);
            need = std::max(need, static_cast<long long>(V.max_m) + V.lds_coop_stage + so.n_waves * 64);
            V.lds_doubles = static_cast<int>(need + 2);
            // the triangular-solve kernels of the split schedule need far less: more of their workgroups fit a CU
            long long const need_solve = std::max(static_cast<long long>(so.n_waves) * V.lds_sslot,
                                                  static_cast<long long>(V.max_m) + V.lds_coop_stage + so.n_waves * 64);
            V.lds_solve_doubles = static_cast<int>(need_solve + 2);
            V.lds_solve_b_doubles = static_cast<int>(std::max(static_cast<long long>(so.n_waves) * V.lds_sslot_b,
                                                              static_cast<long long>(V.max_m) + V.lds_coop_stage + so.n_waves * 64) + 2);
        }
        V.factor_doubles = std::max<long long>(S.factor_doubles, 1);
        V.arena_doubles = std::max<long long>(S.arena_doubles, 1);
        // the LDS caps are fixed now: layout of every front + the assembly lists that go with it
        // top levels that leave most CUs without a workgroup run ONE 16-wavefront workgroup per front (k_m2_factor_top_wide): always in
        // the one-workgroup-per-CU geometry (few instances), and on the under-filled levels near the root of a sweep (fronts x instances
        // <= CUs + 25 %).  Such a workgroup owns its CU's LDS: whole-front layout up to order ~141, chain links continued in LDS.
        {
            bool const wide_knob = env_int0("PHY_ENGINE_HIP_WIDE_TOP", 1) != 0;
            bool const chain_lds = env_int0("PHY_ENGINE_HIP_TOP_CHAIN_LDS", 1) != 0;  // developer knob: 0 = round 2's layout of the top fronts
            long long const whole_cu = h->lds_limit / 8 - 160 - 8;
            for(int l = 0; l < 64; ++l) V.top_wide[l] = (l < V.n_top_levels && wide_knob && (!V.high_occupancy || V.top_cnt[l] * batch <= 320)) ? 1 : 0;
            V.lds_top_doubles = chain_lds ? static_cast<int>(std::max<long long>(V.lds_doubles, whole_cu)) : V.lds_doubles;
            if(!pe::build_assembly_lists(S, V.lds_slot, V.lds_doubles - 2, chain_lds ? V.top_wide : nullptr, V.lds_top_doubles - 2))
                return fail(h, PE_HIP_ERR_INTERNAL, "symbolic analysis: " + S.error);
        }
        HIPCHK(h, pool.upload(V.f_mode, S.f_mode));
        HIPCHK(h, pool.upload(V.f_keep, S.f_keep));
        HIPCHK(h, pool.upload(V.gl_ptr, S.gl_ptr));
        HIPCHK(h, pool.upload(V.gl_rptr, S.gl_rptr));
        HIPCHK(h, pool.upload(V.gl_sptr, S.gl_sptr));
        HIPCHK(h, pool.upload(V.gl_dst, S.gl_dst));
        HIPCHK(h, pool.upload(V.gl_cnt, S.gl_cnt));
        HIPCHK(h, pool.upload(V.gl_src, S.gl_src));
        HIPCHK(h, pool.alloc(V.zero, 1));
        // lane-group kernel of the wave fronts (pe_quad.hpp): its tables; V.q_list / V.n_quads follow the `active` mask (upload_active)
        V.quad = 0;
        V.quad_back = 0;
        V.n_mid = 0;
        if(S.quad)
        {
            HIPCHK(h, pool.upload(V.q_prog, S.q_prog));
            HIPCHK(h, pool.upload(V.q_lists, S.q_lists));
            HIPCHK(h, pool.upload(V.q_lane, S.q_lane));
            HIPCHK(h, pool.upload(V.q_bprog, S.q_bprog));
            HIPCHK(h, pool.upload(V.q2_prog, S.q2_prog));
            HIPCHK(h, pool.upload(V.q2_lists, S.q2_lists));
            HIPCHK(h, pool.upload(V.q2_lane, S.q2_lane));
            HIPCHK(h, pool.upload(V.f_kind, S.f_kind));
            HIPCHK(h, pool.upload(V.f_quad, S.f_quad));
            V.n_mid = S.n_mid;
            V.q_zero_off = S.q_zero_off;
            // LDS stack of a quad: slot 0 of an instance's stack holds a zero, the stride puts the four instances on different banks
            V.q_lds_stride = S.q_lds_doubles > 0 ? (S.q_lds_doubles + 1 + 31) / 32 * 32 + 8 : 0;
            // a quad addresses its four instances by 32-bit byte offsets from the first one: every per-instance array must leave room
            // for at least one instance inside 4 GiB (else the wave fronts fall back to the per-instance path of factor_part)
            long long const stride = 8 * std::max({static_cast<long long>(S.nnzA), V.factor_doubles, V.arena_doubles, static_cast<long long>(S.n)});
            V.quad = stride < (1ll << 31) ? (env_int0("PHY_ENGINE_HIP_QUAD", 1) | 1) : 0;
            V.quad_back = (V.quad && env_int0("PHY_ENGINE_HIP_QUAD_BACK", 1) != 0) ? 1 : 0;  // the same fronts' backward pass on the lane-group kernel
        }
        HIPCHK(h, pool.alloc(V.factor, static_cast<size_t>(V.factor_doubles) * batch));
        HIPCHK(h, pool.alloc(V.arena, static_cast<size_t>(V.arena_doubles) * batch));
        return PE_HIP_OK;
    }

    // launch geometry -> symbolic limits: 8 wavefronts per workgroup, panels / wave slots carved from the LDS limit
    int env_int0(char const* name, int def)
    {
        char const* v = std::getenv(name);
        return v && *v ? std::atoi(v) : def;
    }

    pe::SymbolicOptions symbolic_options(pe_hip_engine const* h, int batch_in, int rows, int panel_reserve = 384, int force_resident = 0)
    {
        int const batch = geometry_batch(batch_in);
        pe::SymbolicOptions so{};
        // Workgroup geometry by batch size (measured on MI355X, profiles/ and scripts/sweep_split_*.sh).
        // Large circuits run the split schedule (one launch per phase): from ~100 instances on, 256-thread workgroups at four per
        // CU with every instance cut into 4 (8, 16) parts -- >= 1024 workgroups for the low-register kernels; fewer instances keep one
        // big workgroup per CU and more parts.  Small circuits run the resident kernel: geometry by the batch alone.
        bool const large = rows >= 3000 && env_int0("PHY_ENGINE_HIP_SPLIT", -1) != 0;
        bool const four_per_cu = large ? batch >= 96 : batch >= 768;
        if(four_per_cu)
        {
            so.n_waves = 4;
            so.wave_m = 45;       // one wavefront takes fronts up to order 45: whole in its 10 KB slot up to 35, the larger ones in the
            so.wave_slot = (pe::pe_ld(35) + 1) * 35;  // panel layout (their panels fit the same slot) -- a third of what used to be
                                  // cooperative fronts leaves the barrier-synchronised phase (-1.3 % per iteration at 1 024 instances, -1.5 % at 128)
            so.wave_p = 16;       // (small staged blocks: the backward kernel of the split schedule then fits 8 workgroups per CU)
            so.absorb_m = 35;
            so.max_pivots = 32;
        }
        else if(batch >= 384)
        {
            so.n_waves = 8;
            so.wave_m = 32;
            so.max_pivots = 32;
        }
        else
        {
            so.n_waves = 8;
            so.wave_m = 56;
            so.wave_p = 20;
            so.max_pivots = 48;
        }
        if(large)
        {
            // (re-swept after the larger wave-front class: 128 instances 16 parts 1.65 ms per iteration against 1.68 with 8 and 1.77 with 12;
            //  256 instances 8 parts 2.74 against 2.81 with 4; 512 and 1 024 instances stay at 4)
            so.n_parts = batch >= 384 ? 4 : (batch >= 192 ? 8 : (batch >= 96 ? 16 : std::clamp(256 / std::max(1, batch), 1, 48)));
            so.part_cut = 1.0;
            so.nd_leaf = 10;  // finer dissection: fewer, better-shaped fronts on big meshes (-3.6 % per iteration on M10k, profiles/sweep_r02_leaf.log);
                              // small circuits keep 24 (their whole graph is one minimum-degree leaf, as validated by every golden)
        }
        // tuning knobs (PHY_ENGINE_HIP_* family, SURVEY.md 5 "Config / flags")
        auto env_int = [](char const* name, int def)
        {
            char const* v = std::getenv(name);
            return v && *v ? std::atoi(v) : def;
        };
        // the wave fronts of a large sweep run four instances per wavefront on the lane-group kernel (pe_quad.hpp): fronts of order
        // <= 32 with <= 16 pivots; larger ones stay with the cooperative phase
        so.quad = (four_per_cu && large && env_int("PHY_ENGINE_HIP_QUAD", 1) != 0) ? 1 : 0;
        if(so.quad)
        {
            // Amalgamation re-swept WITH the lane-group kernel (profiles/sweep_r03_amalgamation.log): a front that absorption grows past
            // order 32 drops out of the quad class, and with it every ancestor inside its wave subtree.  Absorbing only up to order 32 and
            // forcing last-child merges only up to 4 pivots (8 before) leaves 933 of 995 wave fronts to the lane-group kernel on M10k (706
            // of 814 before), 5 % fewer stored factor entries: launch pair -2.3 %, steps/s +1.8 % at 1 024 instances (three interleaved runs).
            // (128 instances -- 16 parts -- do not gain: 31.8 k against 32.0 k steps/s; 256: +2 %.  From 192 instances on.)
            if(batch >= 192)
            {
                so.absorb_m = 32;
                so.relax_small = 4;
            }
            // (the wave-front class keeps the geometry above: wave fronts that do not qualify for the lane-group kernel -- order 33..45, or
            //  above one -- stay with the per-instance wave phase, which is cheaper for them than the cooperative phase)
            so.quad_mid = env_int("PHY_ENGINE_HIP_MID", 0) != 0 ? 1 : 0;  // measured slower than the cooperative phase (pe_quad.hpp): off
            // update matrices whose parent follows in the same list could stay on an LDS stack: 8 wavefronts per CU (two per SIMD at this
            // kernel's register count) share the 160 KB -> 600 doubles per instance of a quad
            // (measured slower than the arena for the fronts it applies to, pe_quad.hpp PE_QUAD_LDS_STACK: off unless asked for)
            so.quad_lds_doubles = std::max(0, env_int("PHY_ENGINE_HIP_QUAD_STACK", 0));
        }
        so.n_waves = std::clamp(env_int("PHY_ENGINE_HIP_WAVES", so.n_waves), 1, PE_THREADS / 64);
        so.wave_m = std::max(1, env_int("PHY_ENGINE_HIP_WAVE_M", so.wave_m));
        so.wave_p = std::max(1, env_int("PHY_ENGINE_HIP_WAVE_P", so.wave_p));
        so.absorb_m = std::max(1, env_int("PHY_ENGINE_HIP_ABSORB_M", so.absorb_m));
        so.nd_leaf = std::max(2, env_int("PHY_ENGINE_HIP_ND_LEAF", so.nd_leaf));
        so.relax_zero_frac = 0.01 * std::clamp(env_int("PHY_ENGINE_HIP_RELAX_X100", static_cast<int>(so.relax_zero_frac * 100.0 + 0.5)), 0, 100);
        so.relax_small = std::max(1, env_int("PHY_ENGINE_HIP_RELAX_SMALL", so.relax_small));
        so.cut_factor = 0.1 * std::max(1, env_int("PHY_ENGINE_HIP_CUT_X10", static_cast<int>(so.cut_factor * 10.0)));
        so.max_pivots = std::clamp(env_int("PHY_ENGINE_HIP_MAX_PIVOTS", so.max_pivots), 1, 64);  // (the triangular solves keep one pivot per lane)
        so.n_parts = std::clamp(env_int("PHY_ENGINE_HIP_PARTS", so.n_parts), 1, 64);
        if(env_int("PHY_ENGINE_HIP_SPLIT", -1) == 0) so.n_parts = 1;  // the resident kernel handles one part per instance
        so.part_cut = 0.1 * std::max(1, env_int("PHY_ENGINE_HIP_PART_CUT_X10", static_cast<int>(so.part_cut * 10.0)));
        so.wave_p = std::min(so.wave_p, so.max_pivots);
        // LDS share of one workgroup: the 128-VGPR kernels keep 16 wavefronts per CU resident (16 / n_waves workgroups)
        bool const shared_cu = four_per_cu || batch >= 384;  // 128-VGPR kernels, 16 wavefronts per CU
        int const resident = force_resident > 0 ? force_resident : std::clamp(env_int("PHY_ENGINE_HIP_RESIDENT", shared_cu ? std::max(1, 16 / so.n_waves) : 1), 1, 8);
        so.shared_cu = (resident > 1) ? 1 : 0;
        long long const lds_doubles = (h->lds_limit / 8 - 160) / resident - 8;  // minus the static LDS of __syncthreads_or & co.
        // a wavefront's slot holds whole fronts of order <= wave_m (pe_front.hpp, FULL mode)
        if(int const ws = env_int("PHY_ENGINE_HIP_WAVE_SLOT", 0); ws > 0) so.wave_slot = ws;  // tuning knob: slot smaller than wave_m needs whole
        if(so.wave_slot > 0 && so.n_waves * so.wave_slot > lds_doubles) so.wave_slot = 0;
        if(so.wave_slot == 0)
            while(static_cast<long long>(so.n_waves) * so.wave_m * (pe::pe_ld(so.wave_m) + 1) > lds_doubles && so.wave_m > 8) --so.wave_m;
        so.wave_p = std::min(so.wave_p, so.wave_m);
        so.absorb_m = std::min(so.absorb_m, so.wave_m);
        // large (panel-mode) fronts keep room behind the panels for the right-hand-side column (m doubles) and their
        // children's staged inverse maps
        so.panel_doubles = std::max<long long>(lds_doubles - panel_reserve, lds_doubles / 2);
        so.panel_reserve = lds_doubles - so.panel_doubles;
        return so;
    }

    // Symbolic analysis + the LDS-fit escalation every caller needs (resident circuit AND the solve_csr_real seam): a front's
    // right-hand-side column (m doubles) must fit the reserve behind its panels, and the top of the tree the launch table.
    // (1) a larger reserve; (2) the whole LDS of a CU for one workgroup; else give up loudly.  `geometry_rows`: row count the
    // launch geometry is chosen by (0: the resident single-workgroup kernel, as the solver seam runs).
    int analyze_fitting(pe_hip_engine* h, int batch, int geometry_rows, int n, int const* rp, int const* ci, double const* vals, pe::Symbolic& S,
                        pe::SymbolicOptions& so)
    {
        so = symbolic_options(h, batch, geometry_rows);
        for(int attempt = 0;; ++attempt)
        {
            if(!pe::analyze(n, rp, ci, vals, so, S))
                return fail(h, S.structurally_singular ? PE_HIP_ERR_SINGULAR : PE_HIP_ERR_INTERNAL, "symbolic analysis: " + S.error);
            bool const too_deep = static_cast<int>(S.top_ptr.size()) - 1 > 64;
            bool const fits = S.max_m + 8 <= so.panel_reserve;
            if(fits && !too_deep) return PE_HIP_OK;
            if(attempt == 2) return fail(h, PE_HIP_ERR_INTERNAL, "symbolic analysis: a front of order " + std::to_string(S.max_m) + " does not fit the LDS of a CU");
            so = symbolic_options(h, batch, geometry_rows, std::max(384, S.max_m + 72), attempt == 1 ? 1 : 0);
            if(too_deep) so.n_parts = 1;
        }
    }

    int ensure_symbolic(pe_hip_engine* h, bool tr, double dt)
    {
        int const cls = tr ? 1 : 0;
        // the static pivot order was matched on representative values at ONE dt (capacitor / inductor companions scale with 1/dt):
        // a time step more than a decade away from it gets a fresh analysis, like a change of class
        bool const dt_moved = tr && h->sym_dt > 0.0 && dt > 0.0 && (dt > 10.0 * h->sym_dt || dt < 0.1 * h->sym_dt) && h->sym_values_override.empty();
        if(h->sym_class == cls && !dt_moved) return PE_HIP_OK;
        if(tr) h->sym_dt = dt;
        auto const t0 = clk::now();
        std::vector<double> av;
        if(!h->sym_values_override.empty()) av = h->sym_values_override;
        else
        {
            pe::estimate_values(h->hc, tr, dt, h->opt.g_min, r_open_of(h), av);
            // test knob: a pivot matching that cannot see magnitudes (every structural entry weighs 1) -- the deliberately bad
            // static order the residual safety net is tested against; a re-match on an instance's own values is not affected
            if(char const* k = std::getenv("PHY_ENGINE_HIP_TEST_BLIND_MATCH"); k && *k == '1') std::fill(av.begin(), av.end(), 1.0);
        }
        pe::SymbolicOptions so{};
        {
            int const rc = analyze_fitting(h, h->hc.batch, h->hc.rows, h->hc.rows, h->hc.rp.data(), h->hc.ci.data(), av.data(), h->sym, so);
            if(rc != PE_HIP_OK)
            {
                h->sym_class = -1;
                return rc;
            }
        }
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
        {
            auto const& S = h->sym;
            int nk[4]{};
            for(int s = 0; s < S.nfronts; ++s) ++nk[S.f_kind[s]];
            std::fprintf(stderr, "[pe_hip] schedule: %d fronts (%d wave, %d cooperative, %d top, %d mid), %d parts, %d top levels\n", S.nfronts, nk[0], nk[1],
                         nk[2], nk[3], S.n_parts, static_cast<int>(S.top_ptr.size()) - 1);
            if(S.quad)
                std::fprintf(stderr, "[pe_hip]   lane-group kernel: LDS stack %d doubles per instance holds %lld of %lld update-matrix doubles of the wave fronts\n",
                             S.q_lds_doubles, S.q_lds_kept, S.q_lds_total);
            if(S.quad)
            {
                std::fprintf(stderr, "[pe_hip]   wave-front lists (fronts):");
                for(size_t L = 0; 2 * L + 1 < S.q_lists.size(); ++L) std::fprintf(stderr, " %d", S.q_lists[2 * L + 1]);
                std::fprintf(stderr, "\n[pe_hip]   MID lists (fronts):");
                for(size_t L = 0; 2 * L + 1 < S.q2_lists.size(); ++L) std::fprintf(stderr, " %d", S.q2_lists[2 * L + 1]);
                std::fprintf(stderr, "\n");
                for(int s = 0; s < S.nfronts; ++s)
                    if(S.f_kind[s] == 3 && dump[1] == '4')
                        std::fprintf(stderr, "[pe_hip]   mid front %d: %dx%d children %d parent %d(kind %d)\n", s, S.f_p[s] + S.f_u[s], S.f_p[s], S.f_child_ptr[s + 1] - S.f_child_ptr[s],
                                     S.f_parent[s], S.f_parent[s] >= 0 ? S.f_kind[S.f_parent[s]] : -1);
            }
            for(int kind = 0; kind < 2; ++kind)
            {
                long long cnt[3]{}, su2[3]{}, spanel[3]{};
                long long const cap = kind == 0 ? std::max<long long>(S.wave_panel_doubles, 1) : so.panel_doubles;
                for(int s = 0; s < S.nfronts; ++s)
                {
                    if(S.f_kind[s] != kind) continue;
                    long long const p = S.f_p[s], u = S.f_u[s], m = p + u;
                    int const nch = S.f_child_ptr[s + 1] - S.f_child_ptr[s];
                    int const mode = m * m <= cap ? 0 : (nch == 1 && S.f_u[S.f_child[S.f_child_ptr[s]]] == m ? 2 : 1);
                    ++cnt[mode];
                    su2[mode] += u * u;
                    spanel[mode] += m * p + p * u;
                }
                std::fprintf(stderr, "[pe_hip]   %s fronts (cap ~%lld doubles): whole %lld (S %lld, panels %lld) | panel+pull %lld (S %lld, panels %lld) | chain link %lld (S %lld, panels %lld)\n",
                             kind == 0 ? "wave" : "cooperative", cap, cnt[0], su2[0], spanel[0], cnt[1], su2[1], spanel[1], cnt[2], su2[2], spanel[2]);
            }
            if(dump[1] == '3')
                for(int s = 0; s < S.nfronts; ++s)
                    if(S.f_kind[s] == 0) std::fprintf(stderr, "[pe_hip]   wave front %d: %dx%d children %d\n", s, S.f_p[s] + S.f_u[s], S.f_p[s], S.f_child_ptr[s + 1] - S.f_child_ptr[s]);
            if(dump[1] == '2')
                for(int s = 0; s < S.nfronts; ++s)
                    if(S.f_kind[s] == 1) std::fprintf(stderr, "[pe_hip]   coop front %d: %dx%d children %d\n", s, S.f_p[s] + S.f_u[s], S.f_p[s], S.f_child_ptr[s + 1] - S.f_child_ptr[s]);
            for(std::size_t l = 0; l + 1 < S.top_ptr.size(); ++l)
            {
                std::fprintf(stderr, "[pe_hip]   top level %zu:", l);
                for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k) std::fprintf(stderr, " %dx%d", S.f_p[S.top_list[k]] + S.f_u[S.top_list[k]], S.f_p[S.top_list[k]]);
                std::fprintf(stderr, "\n");
            }
        }
        h->sym_pool.release();
        int const rc = upload_symbolic(h, h->sym_pool, h->sym, so, h->V, h->hc.batch);
        if(rc != PE_HIP_OK) return rc;
        h->active_dev.clear();  // (the quad list behind the mask depends on this analysis' strides)
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
        {
            auto const& S = h->sym;
            std::fprintf(stderr, "[pe_hip]   top fronts, LDS layout (0 whole, 1 panels, 2 chain link, 3 chain link continued in LDS; * = 16-wavefront level, %d doubles):", h->V.lds_top_doubles);
            for(std::size_t l = 0; l + 1 < S.top_ptr.size(); ++l)
                for(int k = S.top_ptr[l]; k < S.top_ptr[l + 1]; ++k) std::fprintf(stderr, " %d%s", S.f_mode[S.top_list[k]], h->V.top_wide[l] ? "*" : "");
            std::fprintf(stderr, "\n");
        }
        if(char const* dump = std::getenv("PHY_ENGINE_HIP_DUMP_SCHEDULE"); dump && *dump == '1')
            std::fprintf(stderr, "[pe_hip]   LDS plan (doubles): factor %d, solves %d, backward %d (wave slot %d = t %d + stage %d + 64 + stack %d), wave front slot %d\n", h->V.lds_doubles,
                         h->V.lds_solve_doubles, h->V.lds_solve_b_doubles, h->V.lds_sslot, h->V.wave_m, h->V.lds_wave_stage, h->V.lds_sslot - h->V.lds_bstack_off, h->V.lds_slot);
        {
            // The matrix values live in FRONT-ASSEMBLY order on the device: slot e of `aval` is the e-th assembled entry
            // (asm_slot is a permutation of the CSR slots), so a front reads its own entries of A as one contiguous run with no
            // index indirection.  The contribution lists of the stamp are permuted to match.
            auto const& S = h->sym;
            auto const& hc = h->hc;
            size_t const nnz = hc.ci.size();
            std::vector<int> ptr2(nnz + 1, 0), src2;
            src2.reserve(hc.a_src.size());
            for(size_t e = 0; e < nnz; ++e)
            {
                int const slot = S.asm_slot[e];
                src2.insert(src2.end(), hc.a_src.begin() + hc.a_ptr[slot], hc.a_src.begin() + hc.a_ptr[slot + 1]);
                ptr2[e + 1] = static_cast<int>(src2.size());
            }
            if(src2.empty()) src2.push_back(0);
            HIPCHK(h, h->sym_pool.upload(h->V.a_ptr, ptr2));
            HIPCHK(h, h->sym_pool.upload(h->V.a_src, src2));
            // x-dependent slots / rows (pe_front.hpp stamp_dynamic_chunk): Newton iterations after the first stamp only these
            h->V.dyn_a = h->V.dyn_b = nullptr;
            h->V.n_dyn_a = h->V.n_dyn_b = 0;
            if(hc.nonlinear)
            {
                std::vector<char> const dyn = pe::dynamic_dv_mask(hc);
                std::vector<int> da, db;
                for(size_t e = 0; e < nnz; ++e)
                    for(int k = ptr2[e]; k < ptr2[e + 1]; ++k)
                        if(dyn[static_cast<size_t>(src2[k] >> 1)])
                        {
                            da.push_back(static_cast<int>(e));
                            break;
                        }
                for(int r = 0; r < hc.rows; ++r)
                    for(int k = hc.b_ptr[r]; k < hc.b_ptr[r + 1]; ++k)
                        if(dyn[static_cast<size_t>(hc.b_src[k] >> 1)])
                        {
                            db.push_back(r);
                            break;
                        }
                h->V.n_dyn_a = static_cast<int>(da.size());
                h->V.n_dyn_b = static_cast<int>(db.size());
                if(da.empty()) da.push_back(0);
                if(db.empty()) db.push_back(0);
                HIPCHK(h, h->sym_pool.upload(h->V.dyn_a, da));
                HIPCHK(h, h->sym_pool.upload(h->V.dyn_b, db));
            }
            h->V.asm_slot = nullptr;  // identity (pe_front.hpp front_factor); the solve_csr_real seam keeps CSR order + the map
            std::vector<int> slot_e(nnz, 0);  // CSR slot -> position in aval (residual check walks A row by row in original order)
            for(size_t e = 0; e < nnz; ++e) slot_e[S.asm_slot[e]] = static_cast<int>(e);
            HIPCHK(h, h->sym_pool.upload(h->V.slot_e, slot_e));
        }
        h->sym_class = cls;
        h->fact_valid = false;
        h->analyze_ms = ms_since(t0);
        return PE_HIP_OK;
    }

    int collect_stats(pe_hip_engine* h, std::vector<long long> const& steps0, std::vector<long long> const& iters0, pe_hip_run_stats* st)
    {
        int const B = h->hc.batch;
        std::vector<long long> s1(B), i1(B);
        std::vector<int> status(B);
        HIPCHK(h, hipMemcpy(s1.data(), h->V.n_steps, B * sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(i1.data(), h->V.n_iters, B * sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(status.data(), h->V.status, B * sizeof(int), hipMemcpyDeviceToHost));
        int nfail = 0, first = 0;
        long long ds = 0, di = 0;
        for(int b = 0; b < B; ++b)
        {
            ds += s1[b] - steps0[b];
            di += i1[b] - iters0[b];
            if(status[b] != 0)
            {
                if(!nfail) first = status[b];
                ++nfail;
            }
        }
        if(st)
        {
            st->steps = ds;
            st->newton_iters = di;
            st->n_failed = nfail;
        }
        if(nfail)
        {
            h->err = first == PE_HIP_ERR_SINGULAR     ? "singular matrix (zero / non-finite pivot)"
                     : first == PE_HIP_ERR_INACCURATE ? "linear solve left a residual above residual_tol (static pivot order unsuitable for these values)"
                                                      : "Newton iteration did not converge";
            return first;
        }
        return PE_HIP_OK;
    }

    int snapshot_counters(pe_hip_engine* h, std::vector<long long>& s0, std::vector<long long>& i0)
    {
        int const B = h->hc.batch;
        s0.resize(B);
        i0.resize(B);
        HIPCHK(h, hipMemcpy(s0.data(), h->V.n_steps, B * sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(i0.data(), h->V.n_iters, B * sizeof(long long), hipMemcpyDeviceToHost));
        return PE_HIP_OK;
    }
}  // namespace

namespace
{
    // ---------------- multi-workgroup mode: the Newton / TR loops of circult::solve / analyze (circuit.h:892-985, 233-256)
    // driven from the host, one kernel sequence per Newton iteration (pe_kernels.hip: launch_m2_iteration)
    struct M2State
    {
        std::vector<int> status, active, flags;
        std::vector<long long> steps, iters;
        std::vector<double> t;
        std::vector<int> trace;
    };

    int m2_pull(pe_hip_engine* h, M2State& S)
    {
        int const B = h->hc.batch;
        S.status.resize(B);
        S.active.assign(B, 0);
        S.flags.assign(B, 0);
        S.steps.resize(B);
        S.iters.resize(B);
        S.t.resize(B);
        HIPCHK(h, hipMemcpy(S.status.data(), h->V.status, B * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(S.steps.data(), h->V.n_steps, B * sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(S.iters.data(), h->V.n_iters, B * sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(S.t.data(), h->V.t_now, B * sizeof(double), hipMemcpyDeviceToHost));
        return PE_HIP_OK;
    }

    int m2_push(pe_hip_engine* h, M2State const& S, double last_step, bool write_last_step)
    {
        int const B = h->hc.batch;
        HIPCHK(h, hipMemcpy(h->V.status, S.status.data(), B * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->V.n_steps, S.steps.data(), B * sizeof(long long), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->V.n_iters, S.iters.data(), B * sizeof(long long), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->V.t_now, S.t.data(), B * sizeof(double), hipMemcpyHostToDevice));
        if(write_last_step)
        {
            std::vector<double> ls(B, last_step);
            HIPCHK(h, hipMemcpy(h->V.last_step, ls.data(), B * sizeof(double), hipMemcpyHostToDevice));
        }
        if(!S.trace.empty())
        {
            int len = 0;
            HIPCHK(h, hipMemcpy(&len, h->V.trace_len, sizeof(int), hipMemcpyDeviceToHost));
            int const room = std::max(0, h->V.trace_cap - len);
            int const n = std::min<int>(room, static_cast<int>(S.trace.size()));
            if(n > 0) HIPCHK(h, hipMemcpy(h->V.trace + len, S.trace.data(), n * sizeof(int), hipMemcpyHostToDevice));
            len += static_cast<int>(S.trace.size());
            HIPCHK(h, hipMemcpy(h->V.trace_len, &len, sizeof(int), hipMemcpyHostToDevice));
        }
        return PE_HIP_OK;
    }

    int ensure_pinned(pe_hip_engine* h, size_t n)
    {
        if(h->pin_cap >= n) return PE_HIP_OK;
        if(h->pin_active) (void)hipHostFree(h->pin_active);
        if(h->pin_flags) (void)hipHostFree(h->pin_flags);
        h->pin_active = h->pin_flags = nullptr;
        h->pin_cap = 0;
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_active), n * sizeof(int), hipHostMallocDefault));
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_flags), n * sizeof(int), hipHostMallocDefault));
        h->pin_cap = n;
        return PE_HIP_OK;
    }
    // `active` mask of the next launches (stream-ordered).  The caller synchronises the stream before it changes the mask again,
    // so the one pinned staging buffer is free by then.
    // Quad mode: the active instances, ascending, are packed four to a wavefront of the lane-group kernel (pe_quad.hpp) -- behind the
    // mask in the same buffer / the same copy.  A quad addresses its members by 32-bit byte offsets from its first one, so it only
    // takes instances inside that window (a sparse tail of a sweep gives short quads, padded with -1).
    int upload_active(pe_hip_engine* h, std::vector<int> const& mask)
    {
        if(h->active_dev == mask) return PE_HIP_OK;
        size_t const B = mask.size();
        if(int const rc = ensure_pinned(h, 5 * B); rc != PE_HIP_OK) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));  // (an earlier upload from the staging buffer may still be in flight)
        std::copy(mask.begin(), mask.end(), h->pin_active);
        size_t words = B;
        if(h->V.quad)
        {
            long long const stride = 8 * std::max({static_cast<long long>(h->V.nnzA), h->V.factor_doubles, h->V.arena_doubles, static_cast<long long>(h->V.rows)});
            long long const span = std::max<long long>(0, ((1ll << 32) - 1) / std::max<long long>(stride, 1) - 2);
            int* ql = h->pin_active + B;
            int nq = 0, cnt = 0, first = 0;
            for(size_t b = 0; b < B; ++b)
            {
                if(!mask[b]) continue;
                if(cnt == 0 || cnt == 4 || static_cast<long long>(b) - first > span)
                {
                    for(; cnt > 0 && cnt < 4; ++cnt) ql[4 * (nq - 1) + cnt] = -1;
                    ++nq;
                    cnt = 0;
                    first = static_cast<int>(b);
                }
                ql[4 * (nq - 1) + cnt++] = static_cast<int>(b);
            }
            for(; cnt > 0 && cnt < 4; ++cnt) ql[4 * (nq - 1) + cnt] = -1;
            h->V.n_quads = nq;
            h->V.q_list = h->V.active + B;
            words = B + 4 * static_cast<size_t>(nq);
        }
        HIPCHK(h, hipMemcpyAsync(h->V.active, h->pin_active, words * sizeof(int), hipMemcpyHostToDevice, h->stream));
        h->active_dev = mask;
        return PE_HIP_OK;
    }
    int download_flags(pe_hip_engine* h, std::vector<int>& flags)
    {
        if(int const rc = ensure_pinned(h, flags.size()); rc != PE_HIP_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(h->pin_flags, h->V.flags, flags.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::copy(h->pin_flags, h->pin_flags + flags.size(), flags.begin());
        return PE_HIP_OK;
    }

    // Residual safety net on the host-driven schedule.  The iteration just launched left the four norms of every active instance's
    // solve in eta_acc.  Instances above the tolerance get up to two rounds of iterative refinement (launch_m2_refine: active =
    // exactly those); their flags are then the Newton / finiteness bits of the corrected x.  What refinement cannot repair leaves
    // the iteration as PE_HIP_ERR_INACCURATE (the caller re-matches on that instance's values and retries the step).
    int m2_check_residuals(pe_hip_engine* h, M2State& S, std::vector<int>& result, int& n_active)
    {
        int const B = h->hc.batch;
        std::vector<double> eta(static_cast<size_t>(B) * 4);
        auto pull_eta = [&]() -> int
        {
            HIPCHK(h, hipMemcpy(eta.data(), h->V.eta_acc, eta.size() * sizeof(double), hipMemcpyDeviceToHost));
            return PE_HIP_OK;
        };
        auto bad = [&](int b)
        {
            double const* n = &eta[4 * static_cast<size_t>(b)];
            double const den = n[1] * n[2] + n[3];
            double const e = den > 0.0 ? n[0] / den : (n[0] > 0.0 ? INFINITY : 0.0);
            return !(e <= h->V.residual_tol);
        };
        // only iterates about to be accepted were checked on the device (k_m2_residual): nothing to read while every active instance
        // still shows a Newton violation
        bool any = false;
        for(int b = 0; b < B && !any; ++b) any = S.active[b] && !(S.flags[b] & 5) && !(h->hc.nonlinear && (S.flags[b] & 2));
        if(!any) return PE_HIP_OK;
        if(int const rc = pull_eta(); rc != PE_HIP_OK) return rc;
        std::vector<int> todo;
        for(int b = 0; b < B; ++b)
            if(S.active[b] && !(S.flags[b] & 5) && !(h->hc.nonlinear && (S.flags[b] & 2)) && bad(b)) todo.push_back(b);
        if(todo.empty()) return PE_HIP_OK;
        std::vector<int> mask(B);
        for(int round = 0; round < 2 && !todo.empty(); ++round)
        {
            std::fill(mask.begin(), mask.end(), 0);
            for(int b: todo) mask[b] = 1;
            if(int const urc = upload_active(h, mask); urc != PE_HIP_OK) return urc;
            HIPCHK(h, pe::launch_m2_refine(h->stream, h->V));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if(int const rc = pull_eta(); rc != PE_HIP_OK) return rc;
            std::vector<int> fl(B);
            HIPCHK(h, hipMemcpy(fl.data(), h->V.flags, B * sizeof(int), hipMemcpyDeviceToHost));
            std::vector<int> still;
            for(int b: todo)
            {
                S.flags[b] = fl[b];
                if((fl[b] & 5) == 0 && bad(b)) still.push_back(b);
                else
                    ++h->n_refined;
            }
            todo.swap(still);
        }
        for(int b: todo)
        {
            result[b] = PE_HIP_ERR_INACCURATE;
            S.active[b] = 0;
            --n_active;
        }
        return PE_HIP_OK;
    }

    // one solve point of every instance whose status is OK; result[b] = iterations (> 0) or a negative status
    int m2_point(pe_hip_engine* h, M2State& S, int mode, double t, double last_step, bool do_factor, std::vector<int>& result, int& launches)
    {
        int const B = h->hc.batch;
        result.assign(B, 0);
        int n_active = 0;
        for(int b = 0; b < B; ++b)
        {
            S.active[b] = S.status[b] == PE_HIP_OK ? 1 : 0;
            n_active += S.active[b];
        }
        int const max_it = h->hc.nonlinear ? h->V.max_newton : 1;
        for(int it = 0; it < max_it && n_active > 0; ++it)
        {
            if(has_overlay(h))
                if(int const rc = overlay_call_all(h, PE_HIP_OVERLAY_ITERATE, mode, t, last_step, &S.active); rc != PE_HIP_OK) return rc;
            if(int const urc = upload_active(h, S.active); urc != PE_HIP_OK) return urc;
            // (test knob PHY_ENGINE_HIP_FULL_STAMP=1: every iteration stamps everything -- the x-dependent-only path must match it bit for bit)
            static bool const full_stamp = env_int0("PHY_ENGINE_HIP_FULL_STAMP", 0) != 0;
            HIPCHK(h, pe::launch_m2_iteration(h->stream, h->V, mode, t, last_step, do_factor, h->evk0, h->evk1, /*stamp_dynamic=*/it > 0 && !full_stamp));
            ++launches;
            if(int const drc = download_flags(h, S.flags); drc != PE_HIP_OK) return drc;  // (synchronises the stream)
            {
                float kms = 0.f;
                if(hipEventElapsedTime(&kms, h->evk0, h->evk1) == hipSuccess)
                {
                    h->dominant_ms += kms;
                    ++h->dominant_launches;
                }
            }
            if(h->V.residual_tol > 0.0)
                if(int const rrc = m2_check_residuals(h, S, result, n_active); rrc != PE_HIP_OK) return rrc;
            for(int b = 0; b < B; ++b)
            {
                if(!S.active[b]) continue;
                int const f = S.flags[b];
                if(f & 5) result[b] = PE_HIP_ERR_SINGULAR;
                else if(!h->hc.nonlinear || !(f & 2))
                {
                    // circuit.h:950-963: an iterate that passed the Newton test is still subject to the models' check_convergence
                    // hooks -- host-stamped models only (the built-in ones have none); a veto costs one more iteration
                    if(has_overlay(h) && h->hc.nonlinear)
                    {
                        int const crc = overlay_call(h, PE_HIP_OVERLAY_CONVERGED, mode, t, last_step, b);
                        if(crc == PE_HIP_OVERLAY_VETO) continue;
                        if(crc != PE_HIP_OK) return crc;
                    }
                    result[b] = it + 1;
                }
                else
                    continue;
                S.active[b] = 0;
                --n_active;
            }
        }
        for(int b = 0; b < B; ++b)
            if(S.active[b])
            {
                result[b] = PE_HIP_ERR_NO_CONVERGENCE;
                S.active[b] = 0;
            }
        return PE_HIP_OK;
    }

    // `only` != null: a retry of exactly those instances after a rolled-back step -- the companion update of that step has already
    // been applied (update_tr_step precedes the failing solve, circuit.h:246-248), so the first step of the retry skips it
    // `retry`: the first step's companion update has already been applied (see above); false for a plain subset of the instances
    int run_m2_tr(pe_hip_engine* h, double dt, int nsteps, int& launches, std::vector<int> const* only = nullptr, bool retry = true)
    {
        M2State S;
        int rc = m2_pull(h, S);
        if(rc != PE_HIP_OK) return rc;
        int const B = h->hc.batch;
        if(!only)
        {
            // The batch is solved in lockstep at ONE time point per launch sequence (sources are evaluated at that t).  Instances that
            // sit at different time points -- one failed and was rolled back in an earlier call while the others went on -- are
            // therefore run group by group, each at its own t (a revived instance must not be solved at the group's time).
            std::vector<double> ts;
            for(int b = 0; b < B; ++b)
                if(S.status[b] == PE_HIP_OK && std::find(ts.begin(), ts.end(), S.t[b]) == ts.end()) ts.push_back(S.t[b]);
            if(ts.size() > 1)
            {
                for(double const tg: ts)
                {
                    std::vector<int> mask(B, 0);
                    for(int b = 0; b < B; ++b) mask[b] = (S.status[b] == PE_HIP_OK && S.t[b] == tg) ? 1 : 0;
                    if(int const grc = run_m2_tr(h, dt, nsteps, launches, &mask, false); grc != PE_HIP_OK) return grc;
                }
                return PE_HIP_OK;
            }
        }
        bool const skip_first = only && retry;
        bool const may_reuse = !h->hc.nonlinear && !h->opt.refactor_every_solve && !has_overlay(h);  // (overlay values may change every solve)
        std::vector<int> res;
        for(int s = 0; s < nsteps; ++s)
        {
            int alive = 0;
            for(int b = 0; b < B; ++b)
            {
                S.active[b] = (S.status[b] == PE_HIP_OK && (!only || (*only)[b])) ? 1 : 0;
                alive += S.active[b];
            }
            if(!alive) break;
            if(only)  // (a retry of some instances: the others must not be touched by m2_point either)
                for(int b = 0; b < B; ++b)
                    if(!(*only)[b] && S.status[b] == PE_HIP_OK) S.status[b] = -1000;
            if(has_overlay(h) && !(skip_first && s == 0))
                if(int const orc = overlay_call_all(h, PE_HIP_OVERLAY_STEP, PE_HIP_MODE_TR, S.t[0], dt, &S.active); orc != PE_HIP_OK) return orc;
            if(int const urc = upload_active(h, S.active); urc != PE_HIP_OK) return urc;
            if(!(skip_first && s == 0)) HIPCHK(h, pe::launch_m2_companion(h->stream, h->V, dt));
            // every live instance sits at the same time point (same dt, lockstep); take it from the first live one
            double t_prev = 0.0;
            for(int b = 0; b < B; ++b)
                if(S.active[b])
                {
                    t_prev = S.t[b];
                    break;
                }
            double const t = t_prev + dt;
            // linear circuit, same dt as the last factorisation: stamp + triangular solves only
            bool const reuse = may_reuse && h->fact_valid && h->fact_dt == dt;
            rc = m2_point(h, S, PE_HIP_MODE_TR, t, dt, !reuse, res, launches);
            if(rc != PE_HIP_OK) return rc;
            if(may_reuse)
            {
                h->fact_valid = true;
                h->fact_dt = dt;
            }
            for(int b = 0; b < B; ++b)
            {
                if(S.status[b] != PE_HIP_OK || res[b] == 0) continue;
                if(b == 0) S.trace.push_back(res[b]);
                if(res[b] < 0) S.status[b] = res[b];  // the failing step is rolled back: t stays at t_prev (circuit.h:249-253)
                else
                {
                    S.t[b] = t;
                    ++S.steps[b];
                    S.iters[b] += res[b];
                }
            }
        }
        for(int b = 0; b < B; ++b)
            if(S.status[b] == -1000) S.status[b] = PE_HIP_OK;
        return m2_push(h, S, dt, true);
    }

    int run_m2_dc(pe_hip_engine* h, int mode, int& launches, std::vector<int> const* only = nullptr)
    {
        M2State S;
        int rc = m2_pull(h, S);
        if(rc != PE_HIP_OK) return rc;
        int const B = h->hc.batch;
        if(only)
            for(int b = 0; b < B; ++b)
                if(!(*only)[b] && S.status[b] == PE_HIP_OK) S.status[b] = -1000;
        std::vector<double> ls(B);
        HIPCHK(h, hipMemcpy(ls.data(), h->V.last_step, B * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<int> res;
        rc = m2_point(h, S, mode, S.t[0], ls[0], true, res, launches);
        if(rc != PE_HIP_OK) return rc;
        for(int b = 0; b < B; ++b)
        {
            if(S.status[b] != PE_HIP_OK || res[b] == 0) continue;
            if(b == 0) S.trace.push_back(res[b]);
            if(res[b] < 0) S.status[b] = res[b];
            else
                S.iters[b] += res[b];
        }
        for(int b = 0; b < B; ++b)
            if(S.status[b] == -1000) S.status[b] = PE_HIP_OK;
        return m2_push(h, S, 0.0, false);
    }

    // Residual safety net, last resort: instances whose solve stayed inaccurate (status PE_HIP_ERR_INACCURATE, step rolled back).
    // First time: leave the resident kernel for the host-driven schedule, which refines.  After that: a new static pivot order
    // from the values of the first failing instance (the order of load time came from instance 0 at the first dt).  Returns the
    // instances to retry (status cleared) grouped by the steps they still owe, or an empty list when nothing more can be done.
    int prepare_inaccurate_retry(pe_hip_engine* h, bool tr, double dt, int attempt, std::vector<long long> const& steps0, int nsteps,
                                 std::vector<std::pair<int, std::vector<int>>>& groups)
    {
        groups.clear();
        int const B = h->hc.batch;
        std::vector<int> status(B);
        std::vector<long long> s1(B);
        HIPCHK(h, hipMemcpy(status.data(), h->V.status, B * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(s1.data(), h->V.n_steps, B * sizeof(long long), hipMemcpyDeviceToHost));
        // a zero / non-finite pivot under the order matched at load time gets ONE re-match on the instance's own values too (a switch
        // toggled through update_param, a parameter that moved by orders of magnitude): the reference would simply pivot elsewhere
        std::vector<int> failed;
        bool any_inaccurate = false;
        for(int b = 0; b < B; ++b)
            if(status[b] == PE_HIP_ERR_INACCURATE || (status[b] == PE_HIP_ERR_SINGULAR && !h->singular_rematched))
            {
                failed.push_back(b);
                any_inaccurate = any_inaccurate || status[b] == PE_HIP_ERR_INACCURATE;
            }
        if(failed.empty() || attempt >= 2) return PE_HIP_OK;
        if(!any_inaccurate) h->singular_rematched = true;  // (once per resident circuit: a structurally singular system stays singular)
        if(any_inaccurate && attempt == 0 && !h->careful) h->careful = true;
        else
        {
            // re-match on the failing instance's own assembled values (device order = front-assembly order -> CSR slots)
            int const b = failed[0];
            size_t const nnz = h->hc.ci.size();
            std::vector<double> tmp(nnz);
            HIPCHK(h, hipMemcpy(tmp.data(), h->V.aval + static_cast<size_t>(b) * nnz, nnz * sizeof(double), hipMemcpyDeviceToHost));
            h->sym_values_override.assign(nnz, 0.0);
            for(size_t e = 0; e < nnz; ++e)
            {
                double const v = std::fabs(tmp[e]);
                h->sym_values_override[h->sym.asm_slot[e]] = v <= 1.7976931348623157e308 ? v : 1.0;  // (a non-finite entry says nothing about magnitude)
            }
            h->sym_class = -1;
            int const rc = ensure_symbolic(h, tr, dt);
            h->sym_values_override.clear();
            if(rc != PE_HIP_OK) return rc;
            ++h->n_rematched;
        }
        for(int b: failed) status[b] = PE_HIP_OK;
        HIPCHK(h, hipMemcpy(h->V.status, status.data(), B * sizeof(int), hipMemcpyHostToDevice));
        for(int b: failed)
        {
            int const owe = tr ? nsteps - static_cast<int>(s1[b] - steps0[b]) : 1;
            auto it = std::find_if(groups.begin(), groups.end(), [&](auto const& g) { return g.first == owe; });
            if(it == groups.end())
            {
                groups.emplace_back(owe, std::vector<int>(B, 0));
                it = groups.end() - 1;
            }
            it->second[b] = 1;
        }
        return PE_HIP_OK;
    }
}  // namespace

extern "C" {

int pe_hip_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* pe_hip_last_error(pe_hip_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pe_hip_create(int device, pe_hip_engine** out)
{
    if(!out) return PE_HIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    {
        g_create_error = "no HIP device visible: the MI355X engine has no CPU fallback";
        return PE_HIP_ERR_NO_DEVICE;
    }
    if(device < 0 || device >= n)
    {
        g_create_error = "device index out of range";
        return PE_HIP_ERR_ARG;
    }
    auto h = std::make_unique<pe_hip_engine>();
    h->device = device;
    if(hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
       hipEventCreate(&h->ev1) != hipSuccess || hipEventCreate(&h->evk0) != hipSuccess || hipEventCreate(&h->evk1) != hipSuccess)
    {
        g_create_error = "HIP runtime initialisation failed";
        return PE_HIP_ERR_NO_DEVICE;
    }
    int lds = 0;
    if(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0) h->lds_limit = lds;
    // test knob: pretend the LDS is smaller (forces the large-front PANEL / chain paths on small circuits)
    if(char const* v = std::getenv("PHY_ENGINE_HIP_LDS_BYTES"); v && *v) h->lds_limit = std::clamp(std::atoi(v), 8192, h->lds_limit);
    h->opt.refactor_every_solve = 1;
    *out = h.release();
    return PE_HIP_OK;
}

void pe_hip_destroy(pe_hip_engine* h)
{
    if(!h) return;
    (void)hipSetDevice(h->device);
    if(h->ac.d_xacc) (void)hipFree(h->ac.d_xacc);
    if(h->ac.d_b0) (void)hipFree(h->ac.d_b0);
    if(h->ac.d_worst) (void)hipFree(h->ac.d_worst);
    if(h->ac.eng) pe_hip_destroy(h->ac.eng);
    (void)hipStreamSynchronize(h->stream);
    h->circ_pool.release();
    h->sym_pool.release();
    h->csr.pool.release();
    (void)hipEventDestroy(h->ev0);
    (void)hipEventDestroy(h->ev1);
    (void)hipEventDestroy(h->evk0);
    (void)hipEventDestroy(h->evk1);
    (void)hipStreamDestroy(h->stream);
    if(h->pin_active) (void)hipHostFree(h->pin_active);
    if(h->pin_flags) (void)hipHostFree(h->pin_flags);
    delete h;
}

int pe_hip_set_options(pe_hip_engine* h, const pe_hip_options* o)
{
    if(!h || !o) return PE_HIP_ERR_ARG;
    bool const gmin_changed = o->g_min != h->opt.g_min;
    double const r_open_before = r_open_of(h);
    h->opt = *o;
    apply_options(h, h->V);
    if(h->loaded && r_open_of(h) != r_open_before)
    {
        HIPCHK(h, hipSetDevice(h->device));
        std::vector<double> col(h->hc.batch);
        for(auto const& g: h->hc.gen)
            if(g.kind == PE_HIP_SWITCH)
            {
                for(int b = 0; b < h->hc.batch; ++b)
                    (void)pe::gen_static_value(g.kind, &h->hc.gen_par[static_cast<size_t>(b) * h->hc.gen_par_len + g.par], r_open_of(h), col[b]);
                HIPCHK(h, hipMemcpy2D(h->V.dv + g.dv, h->hc.dv_len * sizeof(double), col.data(), sizeof(double), sizeof(double), h->hc.batch,
                                      hipMemcpyHostToDevice));
            }
        h->fact_valid = false;
    }
    if(h->loaded && gmin_changed)
    {
        HIPCHK(h, hipSetDevice(h->device));
        std::vector<double> col(h->hc.batch, o->g_min);
        HIPCHK(h, hipMemcpy2D(h->V.dv + pe::DV_GMIN, h->hc.dv_len * sizeof(double), col.data(), sizeof(double), sizeof(double), h->hc.batch,
                              hipMemcpyHostToDevice));
        h->fact_valid = false;
    }
    return PE_HIP_OK;
}

int pe_hip_set_overlay(pe_hip_engine* h, int n_cells, const int* rows, const int* cols, const double* representative, int n_rhs, const int* rhs_rows,
                       int nonlinear, pe_hip_overlay_fn fn, void* user)
{
    if(!h || n_cells < 0 || n_rhs < 0 || (n_cells > 0 && (!rows || !cols)) || (n_rhs > 0 && !rhs_rows)) return PE_HIP_ERR_ARG;
    if((n_cells > 0 || n_rhs > 0) && !fn) return fail(h, PE_HIP_ERR_ARG, "set_overlay: cells without a callback");
    pe::OverlaySpec next;
    next.rows.assign(rows, rows + n_cells);
    next.cols.assign(cols, cols + n_cells);
    if(representative) next.rep.assign(representative, representative + n_cells);
    next.rhs_rows.assign(rhs_rows, rhs_rows + n_rhs);
    next.nonlinear = nonlinear != 0;
    // the cells are part of the sparsity pattern: a different set invalidates the resident circuit (reload it), as a different
    // set of digital drives does
    if(h->loaded && (next.rows != h->overlay.rows || next.cols != h->overlay.cols || next.rhs_rows != h->overlay.rhs_rows || next.nonlinear != h->overlay.nonlinear))
        h->loaded = false;
    h->overlay = std::move(next);
    h->overlay_fn = fn;
    h->overlay_user = user;
    return PE_HIP_OK;
}

int pe_hip_set_digital_drives(pe_hip_engine* h, int count, const int* node, const double* volt)
{
    if(!h || count < 0 || (count > 0 && (!node || !volt))) return PE_HIP_ERR_ARG;
    if(h->loaded)
    {
        // same drives in the same order: only the voltages change; a different set needs pe_hip_load_circuit again
        bool same = count == h->hc.n_drives;
        for(int k = 0; k < count && same; ++k) same = (node[k] == 0 ? -1 : node[k] - 1) == h->hc.drv_node[k];
        if(same)
        {
            HIPCHK(h, hipSetDevice(h->device));
            for(int k = 0; k < count; ++k)
            {
                std::vector<double> col(h->hc.batch, volt[k]);
                HIPCHK(h, hipMemcpy2D(h->V.dv + h->hc.dv_drv + k, h->hc.dv_len * sizeof(double), col.data(), sizeof(double), sizeof(double),
                                      h->hc.batch, hipMemcpyHostToDevice));
            }
        }
        else
            h->loaded = false;  // the resident circuit no longer matches: every compute entry point refuses until reloaded
    }
    h->drv_node.assign(node, node + count);
    h->drv_volt.assign(volt, volt + count);
    return PE_HIP_OK;
}

int pe_hip_load_circuit(pe_hip_engine* h, int n_nodes, int n_branches, int batch, int n_tables, const pe_hip_device_table* tables)
{
    if(!h || (n_tables > 0 && !tables)) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    h->loaded = false;
    h->sym_class = -1;
    h->fact_valid = false;
    h->circ_pool.release();
    h->stats_scratch = nullptr;
    h->stats_doubles = 0;
    h->active_dev.clear();
    h->singular_rematched = false;
    h->sym_dt = 0.0;
    h->careful = false;  // (a false alarm on the previous circuit must not pin this one to the host-driven schedule)
    h->n_refined = h->n_rematched = 0;
    h->sym_pool.release();
    if(h->ac.eng)
    {
        if(h->ac.d_xacc) (void)hipFree(h->ac.d_xacc);
        if(h->ac.d_b0) (void)hipFree(h->ac.d_b0);
        if(h->ac.d_worst) (void)hipFree(h->ac.d_worst);
        pe_hip_destroy(h->ac.eng);
        h->ac = pe_hip_engine::Ac{};
    }
    if(!pe::build_circuit(n_nodes, n_branches, batch, n_tables, tables, static_cast<int>(h->drv_node.size()), h->drv_node.data(),
                          h->drv_volt.data(), h->hc, h->overlay.empty() ? nullptr : &h->overlay))
        return fail(h, PE_HIP_ERR_ARG, "load_circuit: " + h->hc.error);
    return finish_load(h);
}

}  // extern "C"

namespace
{
// device side of a load: uploads h->hc (topology, contribution lists, parameters) and allocates the per-instance state
int finish_load(pe_hip_engine* h)
{
    auto const& hc = h->hc;
    pe::DevView V{};
    V.rows = hc.rows;
    V.n_nodes = hc.n_nodes;
    V.n_branches = hc.n_branches;
    V.batch = hc.batch;
    V.nnzA = static_cast<int>(hc.ci.size());
    V.dv_len = hc.dv_len;
    V.nR = hc.nR(); V.nC = hc.nC(); V.nL = hc.nL(); V.nVdc = hc.nVdc(); V.nVac = hc.nVac(); V.nIdc = hc.nIdc(); V.nD = hc.nD();
    V.nDrv = hc.n_drives;
    V.nTs = hc.nTs();
    V.nCl = hc.nCl();
    V.nN3 = hc.nN3();
    V.nRl = hc.nRl();
    V.nonlinear = hc.nonlinear ? 1 : 0;
    V.dv_r = hc.dv_r; V.dv_cg = hc.dv_cg; V.dv_ci = hc.dv_ci; V.dv_lr = hc.dv_lr; V.dv_lu = hc.dv_lu; V.dv_vdc = hc.dv_vdc;
    V.dv_vac = hc.dv_vac; V.dv_idc = hc.dv_idc; V.dv_dg = hc.dv_dg; V.dv_di = hc.dv_di; V.dv_drv = hc.dv_drv;
    auto& P = h->circ_pool;
    HIPCHK(h, P.upload(V.c_a, hc.c_a));
    HIPCHK(h, P.upload(V.c_b, hc.c_b));
    HIPCHK(h, P.upload(V.l_a, hc.l_a));
    HIPCHK(h, P.upload(V.l_b, hc.l_b));
    HIPCHK(h, P.upload(V.l_k, hc.l_k));
    HIPCHK(h, P.upload(V.vac_k, hc.vac_k));
    HIPCHK(h, P.upload(V.d_a, hc.d_a));
    HIPCHK(h, P.upload(V.d_c, hc.d_c));
    HIPCHK(h, P.upload(V.a_ptr, hc.a_ptr));
    HIPCHK(h, P.upload(V.a_src, hc.a_src));
    HIPCHK(h, P.upload(V.b_ptr, hc.b_ptr));
    HIPCHK(h, P.upload(V.b_src, hc.b_src));
    HIPCHK(h, P.upload(V.c_cap, hc.c_cap));
    HIPCHK(h, P.upload(V.l_ind, hc.l_ind));
    HIPCHK(h, P.upload(V.vac_par, hc.vac_par));
    HIPCHK(h, P.upload(V.d_par, hc.d_par));
    HIPCHK(h, P.upload(V.ts_kind, hc.ts_kind));
    HIPCHK(h, P.upload(V.ts_dv, hc.ts_dv));
    HIPCHK(h, P.upload(V.ts_par, hc.ts_par));
    HIPCHK(h, P.upload(V.cl_n, hc.cl_n));
    HIPCHK(h, P.upload(V.cl_k, hc.cl_k));
    HIPCHK(h, P.upload(V.cl_dv, hc.cl_dv));
    HIPCHK(h, P.upload(V.cl_par, hc.cl_par));
    HIPCHK(h, P.upload(V.n3_kind, hc.n3_kind));
    HIPCHK(h, P.upload(V.n3_n, hc.n3_n));
    HIPCHK(h, P.upload(V.n3_dv, hc.n3_dv));
    HIPCHK(h, P.upload(V.n3_par, hc.n3_par));
    HIPCHK(h, P.upload(V.rl_n, hc.rl_n));
    HIPCHK(h, P.upload(V.rl_dv, hc.rl_dv));
    HIPCHK(h, P.upload(V.rl_par, hc.rl_par));
    HIPCHK(h, P.alloc(V.rl_engaged, std::max<size_t>(1, static_cast<size_t>(hc.batch) * hc.nRl())));
    size_t const B = static_cast<size_t>(hc.batch);
    HIPCHK(h, P.alloc(V.c_hist, B * hc.nC()));
    HIPCHK(h, P.alloc(V.c_prevg, B * hc.nC()));
    HIPCHK(h, P.alloc(V.d_udlast, B * hc.nD()));
    HIPCHK(h, P.alloc(V.d_geq, B * hc.nD()));
    HIPCHK(h, P.alloc(V.d_hist, B * hc.nD()));
    HIPCHK(h, P.alloc(V.d_prevg, B * hc.nD()));
    HIPCHK(h, P.alloc(V.aval, B * V.nnzA));
    HIPCHK(h, P.alloc(V.rhs, B * hc.rows));
    HIPCHK(h, P.alloc(V.x, B * hc.rows));
    HIPCHK(h, P.alloc(V.xprev, B * hc.rows));
    HIPCHK(h, P.alloc(V.w, B * hc.rows));
    HIPCHK(h, P.alloc(V.t_now, B));
    HIPCHK(h, P.alloc(V.last_step, B));
    HIPCHK(h, P.alloc(V.status, B));
    HIPCHK(h, P.alloc(V.n_steps, B));
    HIPCHK(h, P.alloc(V.n_iters, B));
    V.trace_cap = 1 << 16;
    HIPCHK(h, P.alloc(V.trace, static_cast<size_t>(V.trace_cap)));
    HIPCHK(h, P.alloc(V.trace_len, 1));
    HIPCHK(h, P.alloc(V.prof, B * pe::PE_PROF));
    {
        // scratch of pe_hip_sweep_statistics, allocated with the circuit: a first-call hipMalloc costs milliseconds (7.6 ms measured
        // at 128 instances), the statistics themselves 0.05-0.1 ms
        size_t const need = static_cast<size_t>(stats_chunks(static_cast<int>(B)) + 1) * 4 * hc.rows;
        HIPCHK(h, P.alloc(h->stats_scratch, need, false));
        h->stats_doubles = need;
    }
    HIPCHK(h, P.alloc(V.active, 5 * B));  // the mask + the quad list of the lane-group kernel behind it (upload_active)
    HIPCHK(h, P.alloc(V.flags, B));
    // residual safety net: CSR of A in original order (shared) + per-instance refinement buffers
    HIPCHK(h, P.upload(V.csr_rp, hc.rp));
    HIPCHK(h, P.upload(V.csr_ci, hc.ci));
    HIPCHK(h, P.alloc(V.xsave, B * hc.rows));
    HIPCHK(h, P.alloc(V.rres, B * hc.rows));
    HIPCHK(h, P.alloc(V.eta_acc, B * 4));
    V.slot_e = nullptr;  // (set with the symbolic analysis: aval lives in front-assembly order)
    // static part of dv
    {
        std::vector<double> dv(B * hc.dv_len, 0.0);
        for(size_t b = 0; b < B; ++b)
        {
            double* d = &dv[b * hc.dv_len];
            d[pe::DV_ONE] = 1.0;
            d[pe::DV_GMIN] = h->opt.g_min;
            for(int i = 0; i < hc.nR(); ++i) d[hc.dv_r + i] = hc.r_g[b * hc.nR() + i];
            for(int i = 0; i < hc.nVdc(); ++i) d[hc.dv_vdc + i] = hc.vdc_v[b * hc.nVdc() + i];
            for(int i = 0; i < hc.nIdc(); ++i) d[hc.dv_idc + i] = hc.idc_i[b * hc.nIdc() + i];
            for(int k = 0; k < hc.n_drives; ++k) d[hc.dv_drv + k] = hc.drv_volt[k];
            for(auto const& g: hc.gen)
            {
                double sv;
                if(pe::gen_static_value(g.kind, &hc.gen_par[b * hc.gen_par_len + g.par], r_open_of(h), sv)) d[g.dv] = sv;
            }
        }
        double* ddv{};
        HIPCHK(h, P.alloc(ddv, dv.size(), false));
        HIPCHK(h, hipMemcpy(ddv, dv.data(), dv.size() * sizeof(double), hipMemcpyHostToDevice));
        V.dv = ddv;
    }
    apply_options(h, V);
    h->V = V;
    h->loaded = true;
    return PE_HIP_OK;
}
}  // namespace

extern "C" {

int pe_hip_get_info(pe_hip_engine* h, pe_hip_info* out)
{
    if(!h || !out || !h->loaded) return PE_HIP_ERR_ARG;
    std::memset(out, 0, sizeof(*out));
    auto const& hc = h->hc;
    out->rows = hc.rows;
    out->n_nodes = hc.n_nodes;
    out->n_branches = hc.n_branches;
    out->batch = hc.batch;
    out->nnz_a = static_cast<int>(hc.ci.size());
    out->n_r = hc.nR(); out->n_c = hc.nC(); out->n_l = hc.nL(); out->n_v = hc.nVdc() + hc.nVac() + hc.n_drives; out->n_i = hc.nIdc(); out->n_d = hc.nD();
    out->nonlinear = hc.nonlinear;
    if(h->sym_class >= 0)
    {
        out->nnz_lu = h->sym.nnz_LU;
        out->nnz_lu_stored = h->sym.nnz_LU_stored;
        out->n_fronts = h->sym.nfronts;
        out->max_front = h->sym.max_m;
        out->tree_depth = h->sym.tree_depth;
        out->n_parts = h->V.n_parts;
        out->n_top_levels = h->V.n_top_levels;
        out->n_wavefronts = h->V.n_waves;
        out->lds_bytes = h->V.lds_doubles * 8;
        for(int s = 0; s < h->sym.nfronts; ++s)
        {
            long long const stored = 2LL * h->sym.f_p[s] * h->sym.f_u[s] + static_cast<long long>(h->sym.f_p[s]) * h->sym.f_p[s];
            if(h->sym.f_kind[s] == 0) ++out->n_wave_fronts;
            if(h->V.quad && !h->sym.f_quad.empty() && h->sym.f_quad[s])
            {
                ++out->n_quad_fronts;
                out->nnz_lu_stored_quad += stored;
            }
        }
        for(int s = 0; s < h->sym.nfronts; ++s)
            if(h->sym.f_kind[s] == 2) out->nnz_lu_stored_top += 2LL * h->sym.f_p[s] * h->sym.f_u[s] + static_cast<long long>(h->sym.f_p[s]) * h->sym.f_p[s];
        out->n_row_swaps = h->sym.n_row_swaps;
        out->factor_flops = h->sym.flops;
    }
    out->bytes_per_instance = static_cast<long long>((h->circ_pool.bytes + h->sym_pool.bytes) / std::max(1, hc.batch));
    return PE_HIP_OK;
}

int pe_hip_set_time(pe_hip_engine* h, double t, double last_step)
{
    if(!h || !h->loaded) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<double> a(h->hc.batch, t), b(h->hc.batch, last_step);
    HIPCHK(h, hipMemcpy(h->V.t_now, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->V.last_step, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
    return PE_HIP_OK;
}

int pe_hip_reset(pe_hip_engine* h)
{
    if(!h || !h->loaded) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    auto const& hc = h->hc;
    size_t const B = static_cast<size_t>(hc.batch);
    auto& V = h->V;
    HIPCHK(h, hipMemset(V.x, 0, B * hc.rows * sizeof(double)));
    HIPCHK(h, hipMemset(V.c_hist, 0, std::max<size_t>(1, B * hc.nC()) * sizeof(double)));
    HIPCHK(h, hipMemset(V.c_prevg, 0, std::max<size_t>(1, B * hc.nC()) * sizeof(double)));
    HIPCHK(h, hipMemset(V.d_udlast, 0, std::max<size_t>(1, B * hc.nD()) * sizeof(double)));
    HIPCHK(h, hipMemset(V.d_geq, 0, std::max<size_t>(1, B * hc.nD()) * sizeof(double)));
    HIPCHK(h, hipMemset(V.d_hist, 0, std::max<size_t>(1, B * hc.nD()) * sizeof(double)));
    HIPCHK(h, hipMemset(V.d_prevg, 0, std::max<size_t>(1, B * hc.nD()) * sizeof(double)));
    HIPCHK(h, hipMemset(V.rl_engaged, 0, std::max<size_t>(1, B * hc.nRl()) * sizeof(int)));
    HIPCHK(h, hipMemset(V.t_now, 0, B * sizeof(double)));
    HIPCHK(h, hipMemset(V.last_step, 0, B * sizeof(double)));
    HIPCHK(h, hipMemset(V.status, 0, B * sizeof(int)));
    HIPCHK(h, hipMemset(V.n_steps, 0, B * sizeof(long long)));
    HIPCHK(h, hipMemset(V.n_iters, 0, B * sizeof(long long)));
    HIPCHK(h, hipMemset(V.trace_len, 0, sizeof(int)));
    HIPCHK(h, hipMemset(V.prof, 0, B * pe::PE_PROF * sizeof(long long)));
    h->fact_valid = false;
    return PE_HIP_OK;
}

int pe_hip_analyze_tr(pe_hip_engine* h, double dt, int nsteps, pe_hip_run_stats* st)
{
    if(!h || !h->loaded || nsteps < 0 || !(dt > 0.0)) return h ? fail(h, PE_HIP_ERR_ARG, "analyze_tr: bad arguments or no circuit") : PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if(st) std::memset(st, 0, sizeof(*st));
    h->dominant_ms = 0.0;
    h->dominant_launches = 0;
    if(h->hc.rows == 0 || nsteps == 0) return PE_HIP_OK;
    int rc = ensure_symbolic(h, true, dt);
    if(rc != PE_HIP_OK) return rc;
    // A failed solve is not sticky (circuit.h:242-254: the reference rolls tr_duration back, returns false, and the next
    // analyze() simply tries again from that state -- e.g. after the caller raised g_min): every run starts with all instances live.
    HIPCHK(h, hipMemsetAsync(h->V.status, 0, static_cast<size_t>(h->hc.batch) * sizeof(int), h->stream));
    std::vector<long long> s0, i0;
    rc = snapshot_counters(h, s0, i0);
    if(rc != PE_HIP_OK) return rc;
    bool const may_reuse = !h->hc.nonlinear && !h->opt.refactor_every_solve && !has_overlay(h);
    int const chunk = h->hc.rows > 2000 ? 32 : (h->hc.rows > 200 ? 256 : 2048);
    int launches = 0;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    int done = 0;
    if(split_launch(h))
    {
        rc = run_m2_tr(h, dt, nsteps, launches);
        if(rc != PE_HIP_OK) return rc;
        done = nsteps;
    }
    while(done < nsteps)
    {
        bool const reuse = may_reuse && h->fact_valid && h->fact_dt == dt;
        int const n = reuse || !may_reuse ? std::min(chunk, nsteps - done) : 1;  // first step factors, the rest may reuse
        HIPCHK(h, pe::launch_tr_steps(h->stream, h->V, dt, n, reuse));
        ++launches;
        done += n;
        if(may_reuse)
        {
            h->fact_valid = true;
            h->fact_dt = dt;
        }
    }
    for(int attempt = 0; attempt < 2; ++attempt)
        {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            std::vector<std::pair<int, std::vector<int>>> groups;
            rc = prepare_inaccurate_retry(h, true, dt, attempt, s0, nsteps, groups);
            if(rc != PE_HIP_OK) return rc;
            if(groups.empty()) break;
            for(auto const& g: groups)
            {
                rc = run_m2_tr(h, dt, g.first, launches, &g.second);
                if(rc != PE_HIP_OK) return rc;
            }
        }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if(!may_reuse) h->fact_valid = false;
    rc = collect_stats(h, s0, i0, st);
    if(st)
    {
        st->gpu_ms = ms;
        st->n_launches = launches;
        bool const split = h->dominant_launches > 0;
        st->dominant_ms = split ? h->dominant_ms : ms;
        st->dominant_launches = split ? h->dominant_launches : launches;
    }
    return rc;
}

int pe_hip_analyze_dc(pe_hip_engine* h, int mode, pe_hip_run_stats* st)
{
    if(!h || !h->loaded) return PE_HIP_ERR_ARG;
    if(mode != PE_HIP_MODE_OP && mode != PE_HIP_MODE_DC && mode != PE_HIP_MODE_TROP) return fail(h, PE_HIP_ERR_ARG, "analyze_dc: mode must be OP, DC or TROP");
    HIPCHK(h, hipSetDevice(h->device));
    if(st) std::memset(st, 0, sizeof(*st));
    h->dominant_ms = 0.0;
    h->dominant_launches = 0;
    if(h->hc.rows == 0) return PE_HIP_OK;
    int rc = ensure_symbolic(h, false, 0.0);
    if(rc != PE_HIP_OK) return rc;
    HIPCHK(h, hipMemsetAsync(h->V.status, 0, static_cast<size_t>(h->hc.batch) * sizeof(int), h->stream));  // no sticky failure (see analyze_tr)
    std::vector<long long> s0, i0;
    rc = snapshot_counters(h, s0, i0);
    if(rc != PE_HIP_OK) return rc;
    h->fact_valid = false;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if(split_launch(h))
    {
        int launches = 0;
        rc = run_m2_dc(h, mode, launches);
        if(rc != PE_HIP_OK) return rc;
    }
    else
        HIPCHK(h, pe::launch_dc_point(h->stream, h->V, mode));
    for(int attempt = 0; attempt < 2; ++attempt)
        {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            std::vector<std::pair<int, std::vector<int>>> groups;
            rc = prepare_inaccurate_retry(h, false, 0.0, attempt, s0, 1, groups);
            if(rc != PE_HIP_OK) return rc;
            if(groups.empty()) break;
            int launches = 0;
            for(auto const& g: groups)
            {
                rc = run_m2_dc(h, mode, launches, &g.second);
                if(rc != PE_HIP_OK) return rc;
            }
        }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    rc = collect_stats(h, s0, i0, st);
    if(st)
    {
        st->gpu_ms = ms;
        st->n_launches = 1;
        bool const split = h->dominant_launches > 0;
        st->dominant_ms = split ? h->dominant_ms : ms;
        st->dominant_launches = split ? h->dominant_launches : 1;
    }
    return rc;
}

int pe_hip_get_solution(pe_hip_engine* h, int first, int count, double* x)
{
    if(!h || !h->loaded || !x || first < 0 || count < 0 || first + count > h->hc.batch) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(x, h->V.x + static_cast<size_t>(first) * h->hc.rows, static_cast<size_t>(count) * h->hc.rows * sizeof(double),
                        hipMemcpyDeviceToHost));
    return PE_HIP_OK;
}

int pe_hip_set_solution(pe_hip_engine* h, int first, int count, const double* x)
{
    if(!h || !h->loaded || !x || first < 0 || count < 0 || first + count > h->hc.batch) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->V.x + static_cast<size_t>(first) * h->hc.rows, x, static_cast<size_t>(count) * h->hc.rows * sizeof(double),
                        hipMemcpyHostToDevice));
    return PE_HIP_OK;
}

int pe_hip_get_instance_state(pe_hip_engine* h, int first, int count, int* status, long long* steps, long long* iters, double* t)
{
    if(!h || !h->loaded || first < 0 || count < 0 || first + count > h->hc.batch) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if(status) HIPCHK(h, hipMemcpy(status, h->V.status + first, count * sizeof(int), hipMemcpyDeviceToHost));
    if(steps) HIPCHK(h, hipMemcpy(steps, h->V.n_steps + first, count * sizeof(long long), hipMemcpyDeviceToHost));
    if(iters) HIPCHK(h, hipMemcpy(iters, h->V.n_iters + first, count * sizeof(long long), hipMemcpyDeviceToHost));
    if(t) HIPCHK(h, hipMemcpy(t, h->V.t_now + first, count * sizeof(double), hipMemcpyDeviceToHost));
    return PE_HIP_OK;
}

int pe_hip_get_newton_trace(pe_hip_engine* h, int capacity, int* iters, int* n_out)
{
    if(!h || !h->loaded || !n_out || capacity < 0) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    int len = 0;
    HIPCHK(h, hipMemcpy(&len, h->V.trace_len, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = len;
    int const n = std::min({len, capacity, h->V.trace_cap});
    if(n > 0 && iters) HIPCHK(h, hipMemcpy(iters, h->V.trace, n * sizeof(int), hipMemcpyDeviceToHost));
    return PE_HIP_OK;
}

int pe_hip_get_safety_net_counters(pe_hip_engine* h, long long* refined, long long* rematched, int* careful)
{
    if(!h) return PE_HIP_ERR_ARG;
    if(refined) *refined = h->n_refined;
    if(rematched) *rematched = h->n_rematched;
    if(careful) *careful = h->careful ? 1 : 0;
    return PE_HIP_OK;
}

int pe_hip_measure_hbm_ceiling(pe_hip_engine* h, size_t bytes, int reps, double* gbps)
{
    if(!h || !gbps || bytes < 16 || reps < 1) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    bytes &= ~static_cast<size_t>(15);
    Pool tmp;
    char *a{}, *b{};
    HIPCHK(h, tmp.alloc(a, bytes));
    HIPCHK(h, tmp.alloc(b, bytes));
    HIPCHK(h, pe::launch_stream_copy(h->stream, a, b, bytes));  // warm-up (first touch)
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for(int r = 0; r < reps; ++r) HIPCHK(h, pe::launch_stream_copy(h->stream, a, b, bytes));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *gbps = ms > 0.f ? 2.0 * static_cast<double>(bytes) * reps / (static_cast<double>(ms) * 1e-3) / 1e9 : 0.0;
    return PE_HIP_OK;
}

int pe_hip_sweep_statistics(pe_hip_engine* h, double* out)
{
    if(!h || !h->loaded || !out) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    int const rows = h->hc.rows, B = h->hc.batch;
    if(rows == 0) return PE_HIP_OK;
    int const n_chunks = stats_chunks(B);
    size_t const need = static_cast<size_t>(n_chunks + 1) * 4 * rows;
    if(h->stats_doubles < need)  // scratch kept with the resident circuit (an allocation per call would cost more than the kernels)
    {
        HIPCHK(h, h->circ_pool.alloc(h->stats_scratch, need, false));
        h->stats_doubles = need;
    }
    double* partial = h->stats_scratch;
    double* dev_out = h->stats_scratch + static_cast<size_t>(n_chunks) * 4 * rows;
    HIPCHK(h, pe::launch_sweep_statistics(h->stream, h->V, n_chunks, partial, dev_out));
    HIPCHK(h, hipMemcpyAsync(out, dev_out, static_cast<size_t>(4) * rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PE_HIP_OK;
}

int pe_hip_get_matrix(pe_hip_engine* h, int instance, int* row_ptr, int* col_ind, double* vals, double* rhs)
{
    if(!h || !h->loaded || instance < 0 || instance >= h->hc.batch) return PE_HIP_ERR_ARG;
    auto const& hc = h->hc;
    if(row_ptr) std::copy(hc.rp.begin(), hc.rp.end(), row_ptr);
    if(col_ind) std::copy(hc.ci.begin(), hc.ci.end(), col_ind);
    HIPCHK(h, hipSetDevice(h->device));
    if(vals)
    {
        if(h->sym_class < 0) return fail(h, PE_HIP_ERR_ARG, "get_matrix: no analysis has run yet");
        std::vector<double> tmp(hc.ci.size());  // device order = front-assembly order (ensure_symbolic)
        HIPCHK(h, hipMemcpy(tmp.data(), h->V.aval + static_cast<size_t>(instance) * hc.ci.size(), hc.ci.size() * sizeof(double), hipMemcpyDeviceToHost));
        for(size_t e = 0; e < tmp.size(); ++e) vals[h->sym.asm_slot[e]] = tmp[e];
    }
    if(rhs) HIPCHK(h, hipMemcpy(rhs, h->V.rhs + static_cast<size_t>(instance) * hc.rows, hc.rows * sizeof(double), hipMemcpyDeviceToHost));
    return PE_HIP_OK;
}

int pe_hip_update_param(pe_hip_engine* h, int kind, int index, int column, const double* values, int batched)
{
    if(!h || !h->loaded || !values || index < 0 || column < 0) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    auto& hc = h->hc;
    int const B = hc.batch;
    auto val = [&](int b) { return batched ? values[b] : values[0]; };
    auto put_strided = [&](double* dev_base, size_t stride_doubles, std::vector<double> const& col) -> hipError_t
    { return hipMemcpy2D(dev_base, stride_doubles * sizeof(double), col.data(), sizeof(double), sizeof(double), B, hipMemcpyHostToDevice); };
    std::vector<int> const* map = nullptr;
    switch(kind)
    {
        case PE_HIP_R: map = &hc.map_r; break;
        case PE_HIP_C: map = &hc.map_c; break;
        case PE_HIP_L: map = &hc.map_l; break;
        case PE_HIP_VDC: map = &hc.map_vdc; break;
        case PE_HIP_VAC: map = &hc.map_vac; break;
        case PE_HIP_IDC: map = &hc.map_idc; break;
        case PE_HIP_DIODE: map = &hc.map_d; break;
        default:
            if(kind >= PE_HIP_IAC && kind <= PE_HIP_KIND_MAX) break;
            return fail(h, PE_HIP_ERR_ARG, "update_param: unknown kind");
    }
    if(!map)
    {
        auto const& gm = hc.map_gen[kind];
        if(index >= static_cast<int>(gm.size())) return fail(h, PE_HIP_ERR_ARG, "update_param: index out of range");
        if(column >= pe::gen_ncol(kind)) return PE_HIP_ERR_ARG;
        int const g = gm[index];
        if(g < 0) return PE_HIP_OK;
        auto const& d = hc.gen[g];
        if(kind == PE_HIP_VGEN && column == 0) return fail(h, PE_HIP_ERR_ARG, "update_param: the generator type is fixed at load time");
        std::vector<double> col(B);
        h->fact_valid = false;
        for(int b = 0; b < B; ++b)
        {
            hc.gen_par[static_cast<size_t>(b) * hc.gen_par_len + d.par + column] = val(b);
            pe::gen_derive(hc, g, b);
        }
        double sv;
        if(pe::gen_static_value(kind, &hc.gen_par[d.par], r_open_of(h), sv))
        {
            for(int b = 0; b < B; ++b) (void)pe::gen_static_value(kind, &hc.gen_par[static_cast<size_t>(b) * hc.gen_par_len + d.par], r_open_of(h), col[b]);
            HIPCHK(h, put_strided(h->V.dv + d.dv, hc.dv_len, col));
        }
        else if(kind == PE_HIP_COUPLED_L)
        {
            for(int b = 0; b < B; ++b) col[b] = val(b);
            HIPCHK(h, put_strided(const_cast<double*>(h->V.cl_par) + static_cast<size_t>(d.aux) * 3 + column, static_cast<size_t>(hc.nCl()) * 3, col));
        }
        else if(kind == PE_HIP_RELAY)
        {
            for(int b = 0; b < B; ++b) col[b] = val(b);
            HIPCHK(h, put_strided(const_cast<double*>(h->V.rl_par) + static_cast<size_t>(d.aux) * 2 + column, static_cast<size_t>(hc.nRl()) * 2, col));
        }
        else if(kind >= PE_HIP_NMOS)
        {
            for(int c = 0; c < 3; ++c)  // derived columns (a BJT's Is*Area and N*Ut depend on several raw parameters)
            {
                for(int b = 0; b < B; ++b) col[b] = hc.n3_par[(static_cast<size_t>(b) * hc.nN3() + d.aux) * 3 + c];
                HIPCHK(h, put_strided(const_cast<double*>(h->V.n3_par) + static_cast<size_t>(d.aux) * 3 + c, static_cast<size_t>(hc.nN3()) * 3, col));
            }
        }
        else
        {
            for(int b = 0; b < B; ++b) col[b] = val(b);
            HIPCHK(h, put_strided(const_cast<double*>(h->V.ts_par) + static_cast<size_t>(d.aux) * 8 + column, static_cast<size_t>(hc.nTs()) * 8, col));
        }
        return PE_HIP_OK;
    }
    if(index >= static_cast<int>(map->size())) return fail(h, PE_HIP_ERR_ARG, "update_param: index out of range");
    int const j = (*map)[index];
    if(j < 0) return PE_HIP_OK;  // device with an unconnected pin: nothing resident
    std::vector<double> col(B);
    h->fact_valid = false;
    switch(kind)
    {
        case PE_HIP_R:
            if(column != 0) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b) col[b] = hc.r_g[static_cast<size_t>(b) * hc.nR() + j] = 1.0 / val(b);
            HIPCHK(h, put_strided(h->V.dv + hc.dv_r + j, hc.dv_len, col));
            break;
        case PE_HIP_C:
            if(column != 0) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b) col[b] = hc.c_cap[static_cast<size_t>(b) * hc.nC() + j] = val(b);
            HIPCHK(h, put_strided(const_cast<double*>(h->V.c_cap) + j, hc.nC(), col));
            break;
        case PE_HIP_L:
            if(column != 0) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b) col[b] = hc.l_ind[static_cast<size_t>(b) * hc.nL() + j] = val(b);
            HIPCHK(h, put_strided(const_cast<double*>(h->V.l_ind) + j, hc.nL(), col));
            break;
        case PE_HIP_VDC:
            if(column != 0) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b) col[b] = hc.vdc_v[static_cast<size_t>(b) * hc.nVdc() + j] = val(b);
            HIPCHK(h, put_strided(h->V.dv + hc.dv_vdc + j, hc.dv_len, col));
            break;
        case PE_HIP_IDC:
            if(column != 0) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b) col[b] = hc.idc_i[static_cast<size_t>(b) * hc.nIdc() + j] = val(b);
            HIPCHK(h, put_strided(h->V.dv + hc.dv_idc + j, hc.dv_len, col));
            break;
        case PE_HIP_VAC:
            if(column > 2) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b) col[b] = hc.vac_par[(static_cast<size_t>(b) * hc.nVac() + j) * 3 + column] = val(b);
            HIPCHK(h, put_strided(const_cast<double*>(h->V.vac_par) + static_cast<size_t>(j) * 3 + column, static_cast<size_t>(hc.nVac()) * 3, col));
            break;
        case PE_HIP_DIODE:
        {
            if(column >= PE_HIP_DIODE_NPARAM) return PE_HIP_ERR_ARG;
            for(int b = 0; b < B; ++b)
            {
                double* raw = &hc.d_raw[(static_cast<size_t>(b) * hc.nD() + j) * PE_HIP_DIODE_NPARAM];
                raw[column] = val(b);
                pe::diode_derive(raw, &hc.d_par[(static_cast<size_t>(b) * hc.nD() + j) * pe::DP_NCOL]);
            }
            for(int c = 0; c < pe::DP_NCOL; ++c)
            {
                for(int b = 0; b < B; ++b) col[b] = hc.d_par[(static_cast<size_t>(b) * hc.nD() + j) * pe::DP_NCOL + c];
                HIPCHK(h, put_strided(const_cast<double*>(h->V.d_par) + static_cast<size_t>(j) * pe::DP_NCOL + c, static_cast<size_t>(hc.nD()) * pe::DP_NCOL, col));
            }
            break;
        }
    }
    return PE_HIP_OK;
}

int pe_hip_solve_csr_real(pe_hip_engine* h, int n, int nnz, const int* row_ptr, const int* col_ind, const double* values, const double* b, double* x,
                          int copy_pattern, pe_hip_timings* out)
{
    if(!h || n < 0 || nnz < 0 || !row_ptr || !col_ind || !values || !b || !x) return PE_HIP_ERR_ARG;
    auto const t_total = clk::now();
    pe_hip_timings tm{};
    HIPCHK(h, hipSetDevice(h->device));
    if(n == 0) return PE_HIP_OK;
    auto& C = h->csr;
    if(copy_pattern || !C.have || C.n != n || C.nnz != nnz)
    {
        auto const t0 = clk::now();
        C.have = false;
        C.pool.release();
        pe::SymbolicOptions so{};
        if(int const rc = analyze_fitting(h, 1, 0, n, row_ptr, col_ind, values, C.sym, so); rc != PE_HIP_OK) return rc;
        pe::DevView V{};
        V.rows = n;
        V.n_nodes = n;
        V.batch = 1;
        V.nnzA = nnz;
        int rc = upload_symbolic(h, C.pool, C.sym, so, V, 1);
        if(rc != PE_HIP_OK) return rc;
        HIPCHK(h, C.pool.alloc(V.aval, static_cast<size_t>(nnz)));
        HIPCHK(h, C.pool.alloc(V.rhs, static_cast<size_t>(n)));
        HIPCHK(h, C.pool.alloc(V.x, static_cast<size_t>(n)));
        HIPCHK(h, C.pool.alloc(V.w, static_cast<size_t>(n)));
        HIPCHK(h, C.pool.alloc(V.status, 1));
        C.V = V;
        C.n = n;
        C.nnz = nnz;
        C.have = true;
        tm.analyze_ms = ms_since(t0);
    }
    auto t0 = clk::now();
    HIPCHK(h, hipMemcpyAsync(C.V.aval, values, static_cast<size_t>(nnz) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(C.V.rhs, b, static_cast<size_t>(n) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    tm.h2d_ms = ms_since(t0);
    t0 = clk::now();
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    HIPCHK(h, pe::launch_factor_solve(h->stream, C.V, true));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    tm.solve_ms = ms;
    tm.solve_host_ms = ms_since(t0);
    t0 = clk::now();
    int status = 0;
    HIPCHK(h, hipMemcpy(&status, C.V.status, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(x, C.V.x, static_cast<size_t>(n) * sizeof(double), hipMemcpyDeviceToHost));
    tm.d2h_ms = ms_since(t0);
    tm.total_host_ms = ms_since(t_total);
    if(out) *out = tm;
    if(status != 0) return fail(h, PE_HIP_ERR_SINGULAR, "solve_csr_real: singular matrix (zero / non-finite pivot)");
    return PE_HIP_OK;
}

int pe_hip_analyze_pattern(int n, const int* row_ptr, const int* col_ind, const double* values, pe_hip_info* out)
{
    if(n < 0 || !row_ptr || !col_ind || !out) return PE_HIP_ERR_ARG;
    pe::Symbolic S;
    pe::SymbolicOptions so{};
    std::memset(out, 0, sizeof(*out));
    if(!pe::analyze(n, row_ptr, col_ind, values, so, S)) return S.structurally_singular ? PE_HIP_ERR_SINGULAR : PE_HIP_ERR_INTERNAL;
    out->rows = n;
    out->nnz_a = row_ptr[n];
    out->nnz_lu = S.nnz_LU;
    out->nnz_lu_stored = S.nnz_LU_stored;
    out->n_fronts = S.nfronts;
    out->max_front = S.max_m;
    out->tree_depth = S.tree_depth;
    out->n_row_swaps = S.n_row_swaps;
    out->factor_flops = S.flops;
    out->bytes_per_instance = (S.factor_doubles + S.arena_doubles) * 8;
    return PE_HIP_OK;
}

int pe_hip_analyze_pattern_fronts(int n, const int* row_ptr, const int* col_ind, const double* values, int capacity, int* pivots, int* updates,
                                  int* parent, int* n_fronts)
{
    if(n < 0 || !row_ptr || !col_ind || !n_fronts || capacity < 0) return PE_HIP_ERR_ARG;
    pe::Symbolic S;
    pe::SymbolicOptions so{};
    if(!pe::analyze(n, row_ptr, col_ind, values, so, S)) return S.structurally_singular ? PE_HIP_ERR_SINGULAR : PE_HIP_ERR_INTERNAL;
    *n_fronts = S.nfronts;
    for(int s = 0; s < S.nfronts && s < capacity; ++s)
    {
        if(pivots) pivots[s] = S.f_p[s];
        if(updates) updates[s] = S.f_u[s];
        if(parent) parent[s] = S.f_parent[s];
    }
    return PE_HIP_OK;
}

int pe_hip_get_phase_clocks(pe_hip_engine* h, int instance, long long* ticks8)
{
    if(!h || !h->loaded || !ticks8 || instance < 0 || instance >= h->hc.batch) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(ticks8, h->V.prof + static_cast<size_t>(instance) * pe::PE_PROF, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return PE_HIP_OK;
}

/* ---- checkpoint / resume of the device-resident simulation state (SURVEY.md 5 "checkpoint/resume", 8f rank 4): everything a
 * transient needs to continue bit-exactly -- solution, time, trapezoidal companion histories, junction limiting state, relay
 * contacts, counters, the device value vector -- of every instance, as one flat little-endian blob.  The circuit itself
 * (topology, parameters) is NOT in the blob: load the same circuit first; the header's sizes are checked. */
extern "C++" {
namespace
{
    struct CkHeader
    {
        char magic[8];
        long long rows, batch, nC, nD, nRl, dv_len;
    };
    struct CkPart
    {
        void* ptr;
        size_t bytes;
    };
    std::vector<CkPart> ck_parts(pe_hip_engine* h)
    {
        auto const& hc = h->hc;
        auto& V = h->V;
        size_t const B = static_cast<size_t>(hc.batch);
        return {{V.x, B * hc.rows * sizeof(double)},
                {V.t_now, B * sizeof(double)},
                {V.last_step, B * sizeof(double)},
                {V.status, B * sizeof(int)},
                {V.n_steps, B * sizeof(long long)},
                {V.n_iters, B * sizeof(long long)},
                {V.c_hist, B * hc.nC() * sizeof(double)},
                {V.c_prevg, B * hc.nC() * sizeof(double)},
                {V.d_udlast, B * hc.nD() * sizeof(double)},
                {V.d_geq, B * hc.nD() * sizeof(double)},
                {V.d_hist, B * hc.nD() * sizeof(double)},
                {V.d_prevg, B * hc.nD() * sizeof(double)},
                {V.rl_engaged, B * hc.nRl() * sizeof(int)},
                {V.dv, B * hc.dv_len * sizeof(double)}};
    }
}  // namespace
}  // extern "C++"

int pe_hip_checkpoint_size(pe_hip_engine* h, size_t* bytes)
{
    if(!h || !h->loaded || !bytes) return PE_HIP_ERR_ARG;
    size_t n = sizeof(CkHeader);
    for(auto const& p: ck_parts(h)) n += p.bytes;
    *bytes = n;
    return PE_HIP_OK;
}

int pe_hip_checkpoint_save(pe_hip_engine* h, void* buffer, size_t capacity)
{
    size_t need = 0;
    if(!buffer || pe_hip_checkpoint_size(h, &need) != PE_HIP_OK || capacity < need) return h ? fail(h, PE_HIP_ERR_ARG, "checkpoint_save: buffer too small or no circuit") : PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    auto const& hc = h->hc;
    CkHeader hd{{'P', 'E', 'H', 'I', 'P', 'C', 'K', '1'}, hc.rows, hc.batch, hc.nC(), hc.nD(), hc.nRl(), hc.dv_len};
    char* o = static_cast<char*>(buffer);
    std::memcpy(o, &hd, sizeof(hd));
    o += sizeof(hd);
    for(auto const& p: ck_parts(h))
    {
        if(p.bytes) HIPCHK(h, hipMemcpy(o, p.ptr, p.bytes, hipMemcpyDeviceToHost));
        o += p.bytes;
    }
    return PE_HIP_OK;
}

int pe_hip_checkpoint_load(pe_hip_engine* h, const void* buffer, size_t size)
{
    size_t need = 0;
    if(!buffer || pe_hip_checkpoint_size(h, &need) != PE_HIP_OK || size != need) return h ? fail(h, PE_HIP_ERR_ARG, "checkpoint_load: size does not match the loaded circuit") : PE_HIP_ERR_ARG;
    auto const& hc = h->hc;
    CkHeader hd{};
    std::memcpy(&hd, buffer, sizeof(hd));
    if(std::memcmp(hd.magic, "PEHIPCK1", 8) != 0 || hd.rows != hc.rows || hd.batch != hc.batch || hd.nC != hc.nC() || hd.nD != hc.nD() || hd.nRl != hc.nRl() ||
       hd.dv_len != hc.dv_len)
        return fail(h, PE_HIP_ERR_ARG, "checkpoint_load: the checkpoint belongs to a different circuit");
    HIPCHK(h, hipSetDevice(h->device));
    char const* i = static_cast<char const*>(buffer) + sizeof(hd);
    for(auto const& p: ck_parts(h))
    {
        if(p.bytes) HIPCHK(h, hipMemcpy(p.ptr, i, p.bytes, hipMemcpyHostToDevice));
        i += p.bytes;
    }
    h->fact_valid = false;
    return PE_HIP_OK;
}

/* Small-signal AC at angular frequency omega (circult::solve_once with iterate_ac, run once per sweep point by
 * run_ac_analysis, circuit.h:389-431): complex MNA system of the devices' AC stamps, non-linear devices at their LAST
 * linearisation (run pe_hip_analyze_dc(OP) first, as circuit.h:196-209 / the ACOP case do), solved in real-equivalent form. */
int pe_hip_analyze_ac(pe_hip_engine* h, double omega, pe_hip_run_stats* st)
{
    if(!h || !h->loaded) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if(st) std::memset(st, 0, sizeof(*st));
    auto& hc = h->hc;
    if(hc.rows == 0) return PE_HIP_OK;
    auto& A = h->ac;
    if(!A.built)
    {
        if(!pe::build_ac_circuit(hc, A.circ, has_overlay(h) ? &h->overlay : nullptr)) return fail(h, PE_HIP_ERR_INTERNAL, "analyze_ac: could not build the AC system");
        if(pe_hip_create(h->device, &A.eng) != PE_HIP_OK) return fail(h, PE_HIP_ERR_NO_DEVICE, "analyze_ac: " + std::string(pe_hip_last_error(nullptr)));
        // The right-hand side of the device copy comes from one value slot per row: the host evaluates the sources' lists
        // and, for the refinement steps below, writes residuals there.
        {
            auto& c = A.circ.hc;
            A.b_ptr0 = c.b_ptr;
            A.b_src0 = c.b_src;
            A.rhs0 = c.dv_len;
            c.dv_len += c.rows;
            c.b_ptr.resize(c.rows + 1);
            c.b_src.resize(c.rows);
            for(int r = 0; r <= c.rows; ++r) c.b_ptr[r] = r;
            for(int r = 0; r < c.rows; ++r) c.b_src[r] = (A.rhs0 + r) << 1;
        }
        A.eng->opt = h->opt;
        A.eng->hc = A.circ.hc;
        int const rc = finish_load(A.eng);
        if(rc != PE_HIP_OK) return fail(h, rc, "analyze_ac: " + A.eng->err);
        A.built = true;
        A.sym_omega = -1.0;
    }
    int const B = hc.batch;
    // the linearisation the small-signal stamps refer to
    pe::AcOperatingPoint op;
    op.d_geq.resize(static_cast<size_t>(B) * hc.nD());
    op.dv.resize(static_cast<size_t>(B) * hc.dv_len);
    op.rl_engaged.resize(static_cast<size_t>(B) * hc.nRl());
    if(!op.d_geq.empty()) HIPCHK(h, hipMemcpy(op.d_geq.data(), h->V.d_geq, op.d_geq.size() * sizeof(double), hipMemcpyDeviceToHost));
    if(!op.dv.empty()) HIPCHK(h, hipMemcpy(op.dv.data(), h->V.dv, op.dv.size() * sizeof(double), hipMemcpyDeviceToHost));
    if(!op.rl_engaged.empty()) HIPCHK(h, hipMemcpy(op.rl_engaged.data(), h->V.rl_engaged, op.rl_engaged.size() * sizeof(int), hipMemcpyDeviceToHost));
    if(has_overlay(h))
    {
        if(hc.batch > 1) return fail(h, PE_HIP_ERR_ARG, "analyze_ac: a host-stamp overlay needs batch = 1 for small-signal analysis");
        // host-stamped models: their iterate_ac hooks stamp complex values at this omega around the operating point held in x
        op.ov_a.assign(2 * static_cast<size_t>(hc.n_ov_a), 0.0);
        op.ov_b.assign(2 * static_cast<size_t>(hc.n_ov_b), 0.0);
        h->ov_x.resize(static_cast<size_t>(hc.rows));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(h->ov_x.data(), h->V.x, static_cast<size_t>(hc.rows) * sizeof(double), hipMemcpyDeviceToHost));
        if(h->overlay_fn(h->overlay_user, PE_HIP_OVERLAY_AC, PE_HIP_MODE_OP, omega, 0.0, h->ov_x.data(), op.ov_a.data(), op.ov_b.data()) != 0)
            return fail(h, PE_HIP_ERR_INTERNAL, "analyze_ac: host-stamp overlay: a model's iterate_ac hook failed");
    }
    auto const& ah = A.circ.hc;
    std::vector<double> dv(static_cast<size_t>(B) * ah.dv_len);
    for(int b = 0; b < B; ++b) pe::fill_ac_values(hc, A.circ, op, b, omega, h->opt.g_min, r_open_of(h), &dv[static_cast<size_t>(b) * ah.dv_len]);
    // The pivot order is static (row matching + ordering on representative values): it is (re)made on the values of
    // instance 0 at this frequency when there is none yet, when omega moved more than a decade away from the one it was made
    // for (reactive entries scale with omega), or when a solve with a stale order hits a bad pivot.
    auto analyse_here = [&]()
    {
        int const nnz = static_cast<int>(ah.ci.size());
        A.eng->sym_values_override.assign(nnz, 0.0);
        for(int s = 0; s < nnz; ++s)
        {
            double acc = 0.0;
            for(int e = ah.a_ptr[s]; e < ah.a_ptr[s + 1]; ++e)
            {
                double const v = dv[ah.a_src[e] >> 1];
                acc = (ah.a_src[e] & 1) ? acc - v : acc + v;
            }
            A.eng->sym_values_override[s] = acc;
        }
        A.eng->sym_class = -1;
        A.sym_omega = omega;
    };
    bool const stale = A.sym_omega < 0.0 || (omega == 0.0) != (A.sym_omega == 0.0) ||
                       (omega != 0.0 && (omega > 10.0 * A.sym_omega || omega < 0.1 * A.sym_omega));
    if(stale) analyse_here();
    // the right-hand side of every instance goes into its value slots (the device copy of the system gathers it from there)
    int const R2 = ah.rows;
    auto gather = [&](int const* ptr, int const* src, double const* d, int s)
    {
        double acc = 0.0;
        for(int e = ptr[s]; e < ptr[s + 1]; ++e) acc = (src[e] & 1) ? acc - d[src[e] >> 1] : acc + d[src[e] >> 1];
        return acc;
    };
    for(int b = 0; b < B; ++b)
    {
        double* d = &dv[static_cast<size_t>(b) * ah.dv_len];
        for(int r = 0; r < R2; ++r) d[A.rhs0 + r] = gather(A.b_ptr0.data(), A.b_src0.data(), d, r);
    }
    if(A.d_len != static_cast<size_t>(B) * R2)
    {
        if(A.d_xacc) (void)hipFree(A.d_xacc);
        if(A.d_b0) (void)hipFree(A.d_b0);
        A.d_xacc = A.d_b0 = nullptr;
        A.d_len = static_cast<size_t>(B) * R2;
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&A.d_xacc), A.d_len * sizeof(double)));
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&A.d_b0), A.d_len * sizeof(double)));
        if(!A.d_worst) HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&A.d_worst), sizeof(double)));
    }
    auto solve = [&](bool upload) -> int
    {
        // every AC point is an independent linear solve: no sticky failure state, no history.  A correction solve keeps the device's
        // value vector: its right-hand-side slots hold the residual the kernel before wrote there.
        if(upload) HIPCHK(h, hipMemcpy(A.eng->V.dv, dv.data(), dv.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemset(A.eng->V.status, 0, static_cast<size_t>(B) * sizeof(int)));
        return pe_hip_analyze_dc(A.eng, PE_HIP_MODE_DC, st);
    };
    int rc = solve(true);
    if(rc == PE_HIP_ERR_SINGULAR && A.sym_omega != omega)
    {
        analyse_here();
        rc = solve(true);
    }
    if(rc != PE_HIP_OK) return fail(h, rc, "analyze_ac: " + A.eng->err);
    // Iterative refinement, on the device (the pivot order is static and the real-equivalent form separates the two halves of a
    // complex pivot: entries like r_open = 1e12 next to j omega C leave errors far above rounding): r = b - A x in fp64 from the
    // system as the device assembled it (k_ac_residual), A dx = r with the same pivot order, x += dx (k_ac_accumulate); at most
    // three rounds, stops once the componentwise backward error is at rounding level.  The host reads one double per round.
    hipStream_t const es = A.eng->stream;
    HIPCHK(h, pe::launch_ac_accumulate(es, A.eng->V, A.d_xacc, A.d_b0, true));
    for(int round = 0; round < 3; ++round)
    {
        HIPCHK(h, pe::launch_ac_residual(es, A.eng->V, A.d_xacc, A.d_b0, A.rhs0, A.d_worst));
        double worst = 0.0;
        HIPCHK(h, hipMemcpyAsync(&worst, A.d_worst, sizeof(double), hipMemcpyDeviceToHost, es));
        HIPCHK(h, hipStreamSynchronize(es));
        if(!(worst > 4.0e-16)) break;
        rc = solve(false);
        if(rc != PE_HIP_OK) return fail(h, rc, "analyze_ac (refinement): " + A.eng->err);
        HIPCHK(h, pe::launch_ac_accumulate(es, A.eng->V, A.d_xacc, A.d_b0, false));
    }
    A.x.resize(A.d_len);
    HIPCHK(h, hipMemcpyAsync(A.x.data(), A.d_xacc, A.d_len * sizeof(double), hipMemcpyDeviceToHost, es));
    HIPCHK(h, hipStreamSynchronize(es));
    return PE_HIP_OK;
}

/* complex solution of the last pe_hip_analyze_ac: re / im [count][rows] (node voltage and branch current phasors) */
int pe_hip_get_solution_ac(pe_hip_engine* h, int first, int count, double* re, double* im)
{
    if(!h || !h->loaded || !h->ac.built || !re || !im || first < 0 || count < 0 || first + count > h->hc.batch) return PE_HIP_ERR_ARG;
    int const N = h->hc.rows;
    if(h->ac.x.size() != static_cast<size_t>(h->hc.batch) * 2 * N) return fail(h, PE_HIP_ERR_ARG, "get_solution_ac: no AC solution yet");
    for(int b = 0; b < count; ++b)
    {
        double const* x2 = &h->ac.x[static_cast<size_t>(first + b) * 2 * N];
        std::memcpy(re + static_cast<size_t>(b) * N, x2, N * sizeof(double));
        std::memcpy(im + static_cast<size_t>(b) * N, x2 + N, N * sizeof(double));
    }
    return PE_HIP_OK;
}

/* all PE_PROF slots (pe_device.hpp): the eight above + per-layout breakdown of the cooperative fronts */
int pe_hip_get_phase_clocks_ex(pe_hip_engine* h, int instance, int capacity, long long* ticks, int* n_out)
{
    if(!h || !h->loaded || !ticks || capacity < 0 || instance < 0 || instance >= h->hc.batch) return PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    int const n = std::min(capacity, static_cast<int>(pe::PE_PROF));
    HIPCHK(h, hipMemcpy(ticks, h->V.prof + static_cast<size_t>(instance) * pe::PE_PROF, n * sizeof(long long), hipMemcpyDeviceToHost));
    if(n_out) *n_out = n;
    return PE_HIP_OK;
}

}  // extern "C"
