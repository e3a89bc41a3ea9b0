/* pe_hip.h -- C ABI of the MI355X-native transient engine (libpe_hip.so).
 *
 * This is the drop-in boundary behind Phy-Engine's solver seam.  Plain pointers and sizes only; every
 * function returns 0 on success and a negative pe_hip_status otherwise; no exception crosses the boundary;
 * a handle is thread-compatible (one thread at a time), distinct handles are independent.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   include/phy_engine/circuits/solver/cuda_sparse_lu.h:465-473   cuda_sparse_lu::solve_csr_real(...)
 *        -> pe_hip_solve_csr_real()             (same arguments, host pointers, caller-owned)
 *   include/phy_engine/circuits/solver/cuda_sparse_lu.h:295-312   cuda_sparse_lu::solve_csr / solve_csr_timed on std::complex<double>
 *        -> pe_hip_solve_csr_complex()          (same arguments; the complex arrays as interleaved doubles)
 *   include/phy_engine/circuits/solver/cuda_sparse_lu.h:27-34     struct timings
 *        -> pe_hip_timings
 *   include/phy_engine/circuits/circuit.h:1122-1482               the CUDA branch of circult::solve_once
 *   include/phy_engine/circuits/circuit.h:233-256,363-374,892-985 TR loop / update_tr_step / Newton loop
 *        -> pe_hip_load_circuit() + pe_hip_analyze_tr()/pe_hip_analyze_dc(): the whole per-time-step path
 *           (device stamps, g_min, LU, triangular solves, Newton test, trapezoidal companion update) stays
 *           resident on the GPU; the host only reads node voltages / branch currents back.
 *   include/phy_engine/circuits/circuit.h:63-68,115-121           cuda_solve_policy / cuda_node_threshold
 *        -> pe_hip_device_count() is what `auto_select` consults.
 */
#ifndef PE_HIP_H
#define PE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pe_hip_engine pe_hip_engine; /* opaque */

enum pe_hip_status
{
    PE_HIP_OK = 0,
    PE_HIP_ERR_ARG = -1,       /* bad argument / call order */
    PE_HIP_ERR_NO_DEVICE = -2, /* no HIP device, or the HIP runtime failed */
    PE_HIP_ERR_SINGULAR = -3,  /* zero / non-finite pivot (reference: factorizationIsOk()==false, circuit.h:1517) */
    PE_HIP_ERR_NO_CONVERGENCE = -4, /* Newton exhausted max_iter (reference: solve() returns false, circuit.h:984) */
    PE_HIP_ERR_INTERNAL = -5,
    PE_HIP_ERR_INACCURATE = -6 /* the static-pivot LU left a residual ||Ax - b|| above tolerance that refinement and re-matching could not
                                  repair (the reference pivots partially, Eigen/src/SparseLU/SparseLU.h:464-469); the step is rolled back */
};

/* device kinds of the resident path; parameter columns per device in `params` */
enum pe_hip_kind
{
    PE_HIP_R = 1,    /* nodes a,b        params: r                           (linear/resistance.h:82-110) */
    PE_HIP_C = 2,    /* nodes a,b        params: C                           (linear/capacitor.h:106-155) */
    PE_HIP_L = 3,    /* nodes a,b +branch params: L                          (linear/inductor.h:134-195) */
    PE_HIP_VDC = 4,  /* nodes a,b +branch params: V                          (linear/VDC.h:82-97) */
    PE_HIP_VAC = 5,  /* nodes a,b +branch params: Vp, omega[rad/s], phase[rad] (linear/VAC.h:162-179) */
    PE_HIP_IDC = 6,  /* nodes a,b        params: I                           (linear/IDC.h:84-95) */
    PE_HIP_DIODE = 7,/* nodes a,c        params: Is,N,Isr,Nr,Temp,Ibv,Bv,Bv_set,Area,tt, tt_in_tr
                                                                            (non-linear/PN_junction.h:296-503;
                                                                             tt_in_tr=0 for the diodes of a
                                                                             full_bridge_rectifier, which has no
                                                                             iterate_tr: base.h:248-264) */
    /* ---- the remaining linear stampers.  Four-pin kinds take nodes [count][4], kinds with two
     * branches take branch [count][2]; pins in the reference's pin order. */
    PE_HIP_IAC = 8,    /* nodes a,b           params: Ip, omega[rad/s], phase[rad]  TR/TROP only, nothing in OP/DC (linear/IAC.h:124-160) */
    PE_HIP_VCCS = 9,   /* nodes S,T,P,Q       params: g       I(S->T) = g (vP - vQ)                         (linear/VCCS.h:80-95) */
    PE_HIP_VCVS = 10,  /* nodes S,T,P,Q +1 br params: mu      vS - vT = mu (vP - vQ)                        (linear/VCVS.h:81-102) */
    PE_HIP_CCCS = 11,  /* nodes S,T,P,Q +1 br params: alpha   sense branch P->Q (a short), I(S->T) = alpha i (linear/CCCS.h:81-100) */
    PE_HIP_CCVS = 12,  /* nodes S,T,P,Q +2 br params: r       branches: output k, sense c; vS - vT = r i_c  (linear/CCVS.h:80-106) */
    PE_HIP_OPAMP = 13, /* nodes S,T,P,Q +1 br params: mu      output P,Q driven by mu (vS - vT)             (linear/op_amp.h:64-83) */
    PE_HIP_XFMR = 14,  /* nodes P,Q,S,T +2 br params: n       ideal transformer Vp = n Vs, Is = -n Ip       (linear/transformer.h:67-98) */
    PE_HIP_SWITCH = 15,/* nodes a,b     +1 br params: cut_through (0/1): D = -(cut ? 0 : r_open)            (controller/switch.h:86-103) */
    PE_HIP_VGEN = 16,  /* nodes +,-     +1 br params: type, Vh, Vl, freq[Hz], duty, phase[rad], tr, tf
                                              type 0 sawtooth, 1 square, 2 pulse, 3 triangle; OP/DC/TROP take t = 0
                                              (generator/sawtooth.h:88-107, square.h:93-110, pulse.h:107-141, triangle.h:88-112) */
    PE_HIP_COUPLED_L = 17,/* nodes p1,p2,s1,s2 +2 br params: L1, L2, k  trapezoidal 2x2 Thevenin companion in TR, two
                                              shorts otherwise (linear/coupled_inductors.h:92-115,160-246) */
    /* three-pin non-linear devices: nodes [count][3]; re-linearised every Newton iteration on the device */
    PE_HIP_NMOS = 18,     /* nodes D,G,S  params: Kp, lambda, Vth   Shichman-Hodges level 1 (non-linear/nmosfet.h:84-141) */
    PE_HIP_PMOS = 19,     /* nodes D,G,S  params: Kp, lambda, Vth                          (non-linear/pmosfet.h:84-139) */
    PE_HIP_BJT_NPN = 20,  /* nodes B,C,E  params: Is, N, BetaF, Temp, Area   forward-active Ebers-Moll (non-linear/BJT_NPN.h:100-158) */
    PE_HIP_BJT_PNP = 21,  /* nodes B,C,E  params: Is, N, BetaF, Temp, Area                  (non-linear/BJT_PNP.h:100-158) */
    PE_HIP_RELAY = 22,    /* nodes C+,C-,A,B +1 br  params: Von, Voff   contact A-B closes at v(C+)-v(C-) >= Von, opens at <= Voff (state
                             per instance, re-evaluated at every stamp; open = r_open); counts as non-linear (controller/relay.h:75-105) */
    PE_HIP_XFMR_CT = 23   /* nodes P,Q,S1,CT,S2 [count][5] +3 br (kP, kH1, kH2)  params: n_total = Vp / V(S1-S2)
                                                                                   (linear/transformer_center_tap.h:71-132) */
};
#define PE_HIP_DIODE_NPARAM 11
#define PE_HIP_VGEN_NPARAM 8
#define PE_HIP_KIND_MAX 23

/* analysis modes (phy_engine::analyze_type, circuits/analyze.h:7-16) */
enum pe_hip_mode
{
    PE_HIP_MODE_OP = 0,
    PE_HIP_MODE_DC = 1,
    PE_HIP_MODE_TR = 4,
    PE_HIP_MODE_TROP = 5
};

typedef struct pe_hip_device_table
{
    int kind;             /* pe_hip_kind */
    int count;            /* devices in this table */
    const int* nodes;     /* [count][pins] node ids: 0 = ground, 1..n_nodes, -1 = unconnected pin (pins = 2 .. 5: see pe_hip_kind) */
    const int* branch;    /* [count][branches] global branch index (0-based, after digital drives) for kinds with branch rows, else NULL */
    const double* params; /* [batch][count][ncol] when params_batched, else [count][ncol] (shared by every instance) */
    int params_batched;
} pe_hip_device_table;

/* Newton / environment knobs (phy_engine::environment, circuits/environment/environment.h:7-22; defaults of
 * circuit.h:898-903 apply where a field is <= 0) */
typedef struct pe_hip_options
{
    double v_abstol, v_reltol, i_abstol, i_reltol;
    double g_min;
    int max_newton; /* 0 -> 64 */
    int refactor_every_solve; /* 1 (default): factor on every solve_once like the reference; 0: reuse the factors of a linear circuit while dt is unchanged */
    double r_open; /* contact resistance of an open switch (environment.h r_open; <= 0 -> 1e12, circuit.h:1012) */
    double residual_tol; /* safety net of the static-pivot LU: after every linear solve eta = ||Ax - b||_inf / (||A||_inf ||x||_inf + ||b||_inf)
                            is checked per instance; above this (0 -> 1e-10, < 0 disables the check) the solve is refined with the same
                            factors' order, then re-matched on that instance's values, else PE_HIP_ERR_INACCURATE */
} pe_hip_options;

/* mirrors cuda_sparse_lu::timings (cuda_sparse_lu.h:27-34) */
typedef struct pe_hip_timings
{
    double h2d_ms, solve_ms, d2h_ms, solve_host_ms, total_host_ms, analyze_ms;
} pe_hip_timings;

typedef struct pe_hip_info
{
    int rows, n_nodes, n_branches, batch;
    int nnz_a;
    long long nnz_lu;        /* structural nnz(L)+nnz(U) of this engine's ordering (F of SURVEY.md 8d) */
    long long nnz_lu_stored; /* entries held in the dense front panels (incl. relaxation zeros) */
    int n_fronts, max_front, tree_depth, n_row_swaps;
    double factor_flops;
    long long bytes_per_instance; /* resident HBM bytes per circuit instance */
    int n_r, n_c, n_l, n_v, n_i, n_d;
    int nonlinear;
    int n_parts;      /* > 1: multi-workgroup schedule (one circuit spread over n_parts workgroups + top levels) */
    int n_top_levels; /* launches of the top of the tree in that schedule */
    int n_wavefronts; /* wavefronts per workgroup of the launch geometry chosen for this batch */
    int lds_bytes;    /* dynamic LDS per workgroup */
    long long nnz_lu_stored_top; /* part of nnz_lu_stored held by the fronts of the top levels (split schedule: k_m2_factor_top / k_m2_solve_top) */
    int n_wave_fronts;           /* fronts below the cooperative part of the tree (one wavefront each) ... */
    int n_quad_fronts;           /* ... of which the lane-group kernel k_m2_factor_quads factors (four instances per wavefront); 0: not in use */
    long long nnz_lu_stored_quad; /* part of nnz_lu_stored held by those fronts */
    int mid_top_limit, ew_grid, quad_lds_pad; /* launch-shape knobs MID_TOP / EW_GRID / QUAD_LDS in effect for THIS engine's resident circuit
                                                 (pe_hip_set_knob, else the environment, else the default; 0 = rule of the policy) */
} pe_hip_info;

typedef struct pe_hip_run_stats
{
    long long steps;        /* accepted time points, summed over instances */
    long long newton_iters; /* solve_once-equivalents, summed over instances */
    double gpu_ms;          /* HIP-event time of the kernels of this call, on the engine's stream */
    int n_launches;
    int n_failed;           /* instances that stopped early */
    double dominant_ms;     /* HIP-event time of the launches of the dominant kernel within gpu_ms: the resident kernel itself, or in the
                               split schedule k_m2_factor_parts (k_m2_solve_parts when the factors are reused) */
    int dominant_launches;
} pe_hip_run_stats;

int pe_hip_device_count(void);
int pe_hip_create(int device, pe_hip_engine** out);
void pe_hip_destroy(pe_hip_engine* h);
const char* pe_hip_last_error(pe_hip_engine* h); /* valid until the next call on h; h may be NULL (creation errors) */

/* ---- drop-in for cuda_sparse_lu::solve_csr_real (cuda_sparse_lu.h:465-473): A x = b, CSR, sorted columns.
 * copy_pattern != 0: (re)analyse the pattern; == 0: reuse the cached analysis (same n/nnz/pattern). */
int pe_hip_solve_csr_real(pe_hip_engine* h, int n, int nnz, const int* row_ptr, const int* col_ind, const double* values,
                          const double* b, double* x, int copy_pattern, pe_hip_timings* out);

/* ---- drop-in for the complex twin cuda_sparse_lu::solve_csr_timed / solve_csr on std::complex<double> (cuda_sparse_lu.h:295-312), which
 * circult::solve_once calls when the stamped system is not all-real (circuit.h:1332: AC / ACOP).  values_re_im / b_re_im / x_re_im are
 * the arrays of std::complex<double> the reference passes, seen as interleaved (re, im) doubles: 2 nnz, 2 n and 2 n of them.  Solved in
 * real-equivalent form [Ar -Ai; Ai Ar] by the kernels of the real seam + fp64 iterative refinement on the device; copy_pattern as above
 * (a cached pattern keeps its pivot order and is re-analysed once on the current values if a solve with it fails).  Returns
 * PE_HIP_ERR_SINGULAR / PE_HIP_ERR_INACCURATE where the reference's solver returns false. */
int pe_hip_solve_csr_complex(pe_hip_engine* h, int n, int nnz, const int* row_ptr, const int* col_ind, const double* values_re_im,
                             const double* b_re_im, double* x_re_im, int copy_pattern, pe_hip_timings* out);

/* Build id of this library: 16 hex digits of a sha256 over its sources, headers and compile flags (csrc/Makefile).  bench.py prints it and
 * every profile summary under profiles/ carries it, so a figure can be tied to the library that produced it. */
const char* pe_hip_build_id(void);

/* ---- resident path */
int pe_hip_load_circuit(pe_hip_engine* h, int n_nodes, int n_branches, int batch, int n_tables, const pe_hip_device_table* tables);
int pe_hip_set_options(pe_hip_engine* h, const pe_hip_options* opt);
int pe_hip_get_info(pe_hip_engine* h, pe_hip_info* out);

/* Tuning knobs of ONE engine: the launch-geometry / symbolic-analysis parameters INTEGRATION.md lists as the PHY_ENGINE_HIP_* environment
 * family (the counterpart of the reference's cuda_policy / cuda_node_threshold members plus its PHY_ENGINE_CUDA_* variables,
 * circuit.h:63-68, benchmark/README.md:11-21), set per engine instead of per process: `name` with or without the PHY_ENGINE_HIP_
 * prefix ("PARTS", "ABSORB_M", "SPLIT", ...).  A knob set here wins over the environment variable of the same name, which wins over the
 * measured default; it takes effect at the next analysis of the resident circuit (the symbolic analysis is redone).  Test-only
 * variables (…_TEST_*, …_FULL_STAMP, …_DUMP_SCHEDULE, …_LDS_BYTES) stay environment-only. */
int pe_hip_set_knob(pe_hip_engine* h, const char* name, int value);
int pe_hip_get_knob(pe_hip_engine* h, const char* name, int* value, int* is_set); /* what is set (engine, else environment); is_set may be NULL */

/* digital_out of circult (circuit.h:102,509,1015-1022): ideal sources occupying the FIRST `count` branches.
 * Pass the drives before pe_hip_load_circuit() (they are part of the branch numbering).  On a loaded engine the same
 * drive set with new voltages updates in place; a different set invalidates the resident circuit (reload it). */
int pe_hip_set_digital_drives(pe_hip_engine* h, int count, const int* node, const double* volt);

/* ---- Host-stamp overlay: plug-in models WITHOUT a device table.
 * The reference's extension mechanism is the per-model stamp hook (iterate_{dc,tr,op,trop}_define(tag, M&, MNA&[, t]) with the
 * fallback chains of model/model_refs/base.h:216-304, called in the model loop of circult::solve_once, circuit.h:1071-1084).
 * A user model that only has those hooks is evaluated ON THE HOST, once per Newton iteration, into a fixed set of matrix
 * cells / right-hand-side rows discovered once (what mna_keep_pattern_ready does, circuit.h:993-1003); the values are uploaded
 * and ADDED to the device-side stamp.  The hooks read node voltages, so the Newton loop of such a circuit is driven from the
 * host (one callback + one small upload per iteration and instance).  In a batch every group of calls is preceded by
 * PE_HIP_OVERLAY_INSTANCE with the instance index in `mode` (see below) -- small-signal AC included (one PE_HIP_OVERLAY_AC call per instance).
 *   rows / cols / rhs_rows  absolute MNA indices, 0-based: nodes 0 .. n_nodes-1, then branches (mna.h:60-157 G/B/C/D/I/E layout)
 *   representative          |value| per cell for the static pivot matching (the discovery stamp), may be NULL
 *   nonlinear               1: the circuit needs Newton iterations even without a built-in non-linear device
 *   fn(user, event, mode, t, dt, x, a_values, b_values) -> 0 ok, else the analysis fails with PE_HIP_ERR_INTERNAL:
 *     PE_HIP_OVERLAY_STEP     start of a transient step, x = solution of the previous time point, dt = new step (the models'
 *                             step_changed_tr hooks; circult::update_tr_step, circuit.h:363-374); a_values / b_values NULL
 *     PE_HIP_OVERLAY_ITERATE  before the stamp of every Newton iteration, x = current iterate: fill a_values[n_cells] and
 *                             b_values[n_rhs] (mode = pe_hip_mode of the solve, t = time of the point being solved)
 * Call before pe_hip_load_circuit() (the cells are part of the sparsity pattern); n_cells = n_rhs = 0 with fn = NULL removes it. */
#define PE_HIP_OVERLAY_STEP 0
#define PE_HIP_OVERLAY_ITERATE 1
/*     PE_HIP_OVERLAY_CONVERGED  the iterate x has passed the engine's Newton test: the models' check_convergence hooks are consulted
 *                             as circult::solve does (circuit.h:950-963) -- return 0 to accept it, PE_HIP_OVERLAY_VETO to iterate again
 *                             (counts against max_newton like any other iteration); a_values / b_values NULL */
#define PE_HIP_OVERLAY_CONVERGED 2
#define PE_HIP_OVERLAY_VETO 100 /* a RETURN value (of the CONVERGED event only), deliberately unlike every event number and every small error code */
/* COMPATIBILITY NOTE for callbacks written against round 2 (events STEP / ITERATE only): since round 3 every callback of a non-linear
 * circuit also receives PE_HIP_OVERLAY_CONVERGED, every callback of a batch > 1 PE_HIP_OVERLAY_INSTANCE -- both with a_values / b_values
 * NULL -- and PE_HIP_OVERLAY_AC where small-signal analysis is used.  A callback must dispatch on `event` and return 0 for events it does
 * not handle; one that treats "anything but STEP" as ITERATE would write through NULL. */
/*     PE_HIP_OVERLAY_AC       one small-signal point of pe_hip_analyze_ac (the models' iterate_ac hooks, circuit.h:389-431): t carries
 *                             omega, x the operating point; a_values holds 2 n_cells doubles -- the real parts of the cells, then the
 *                             imaginary parts -- and b_values 2 n_rhs likewise */
#define PE_HIP_OVERLAY_AC 3
/*     PE_HIP_OVERLAY_INSTANCE  batch > 1 only, before each of the calls above: they concern instance `mode` of the batch (x, a_values,
 *                             b_values NULL).  Models with state of their own (a junction's last voltage, a companion history) keep one
 *                             copy per instance and switch here; a callback that cannot returns non-zero and the analysis fails. */
#define PE_HIP_OVERLAY_INSTANCE 4
typedef int (*pe_hip_overlay_fn)(void* user, int event, int mode, double t, double dt, const double* x, double* a_values, double* b_values);
int pe_hip_set_overlay(pe_hip_engine* h, int n_cells, const int* rows, const int* cols, const double* representative, int n_rhs, const int* rhs_rows,
                       int nonlinear, pe_hip_overlay_fn fn, void* user);

/* overwrite one parameter column of one device for every instance (values: [batch] if batched else [1]) */
int pe_hip_update_param(pe_hip_engine* h, int kind, int index, int column, const double* values, int batched);

/* tr_duration / last_step of every instance (circuit.h:161-162), e.g. when a netlist is re-loaded mid-simulation */
int pe_hip_set_time(pe_hip_engine* h, double t, double last_step);
int pe_hip_reset(pe_hip_engine* h); /* circult::reset(), circuit.h:446-465: t = 0, x = 0, companion state cleared */

/* one OP / DC / TROP solve (Newton inside), every instance */
int pe_hip_analyze_dc(pe_hip_engine* h, int mode, pe_hip_run_stats* stats);
/* `nsteps` fixed-dt transient steps, every instance: update_tr_step -> t += dt -> Newton(solve_once) */
int pe_hip_analyze_tr(pe_hip_engine* h, double dt, int nsteps, pe_hip_run_stats* stats);

/* Checkpoint / resume of the device-resident simulation state of every instance (solution, time, companion histories,
 * junction and relay state, counters, device values): a transient continued from a loaded checkpoint is bit-identical to an
 * uninterrupted one.  The blob does not contain the circuit: load the same circuit (same tables, same batch) first. */
int pe_hip_checkpoint_size(pe_hip_engine* h, size_t* bytes);
int pe_hip_checkpoint_save(pe_hip_engine* h, void* buffer, size_t capacity);
int pe_hip_checkpoint_load(pe_hip_engine* h, const void* buffer, size_t size);

/* Small-signal AC at angular frequency omega [rad/s] (circult::solve_once with the models' iterate_ac hooks; one call per
 * sweep point of run_ac_analysis, circuit.h:389-431).  Non-linear devices are stamped at their last linearisation: run
 * pe_hip_analyze_dc(PE_HIP_MODE_OP) first, as the reference's AC / ACOP cases do (circuit.h:192-232).  The complex system
 * is solved in real-equivalent form [Ar -Ai; Ai Ar] by the same kernels.  pe_hip_get_solution_ac returns the phasors. */
int pe_hip_analyze_ac(pe_hip_engine* h, double omega, pe_hip_run_stats* stats);
int pe_hip_get_solution_ac(pe_hip_engine* h, int first_instance, int count, double* re, double* im);

/* x = [node voltages ; branch currents], instance-major [count][rows] */
int pe_hip_get_solution(pe_hip_engine* h, int first_instance, int count, double* x);
int pe_hip_set_solution(pe_hip_engine* h, int first_instance, int count, const double* x);
/* per-instance: status (pe_hip_status), accepted steps, Newton iterations, current time */
/* Diagnostics of the residual safety net (pe_hip_options.residual_tol): solves repaired by iterative refinement, symbolic
 * re-analyses on a failing instance's own values, and whether the engine has left the resident kernel for the host-driven
 * (refining) schedule.  Any pointer may be NULL. */
int pe_hip_get_safety_net_counters(pe_hip_engine* h, long long* refined, long long* rematched, int* careful);

/* On-box achievable HBM bandwidth (SURVEY.md 8d "measure the achievable ceiling on the box with a device-to-device stream
 * kernel"): copies `bytes` (two temporary buffers of that size) `reps` times; *gbps = (read + written bytes) / HIP-event time. */
int pe_hip_measure_hbm_ceiling(pe_hip_engine* h, size_t bytes, int reps, double* gbps);

/* The sweep's one exchange step: per-row statistics of the current solution over this engine's instances,
 * computed on the device -- out[0][r] = sum_b x_b[r], out[1][r] = sum_b x_b[r]^2, out[2][r] = min_b, out[3][r] = max_b
 * (out: [4][rows] doubles, host memory).  Ranks combine them with one SUM and one MAX all-reduce (min travels as -min). */
int pe_hip_sweep_statistics(pe_hip_engine* h, double* out);

int pe_hip_get_instance_state(pe_hip_engine* h, int first_instance, int count, int* status, long long* steps, long long* iters, double* t);

/* ---- Monte-Carlo / parameter sweep over several GPUs of one node (csrc/pe_sweep.cpp).
 * Independent instances of ONE topology are the natural shard of this path: the symbolic analysis is replicated per device, the
 * instances are dealt out in contiguous blocks of ceil(batch / G) per device -- the chunk rule of the reference's only
 * multi-device code, src/pe_synth_cuda_u64_cones.cu:1894-1904 -- and, like that code's entry points (:1861-1872: extern "C",
 * device mask first), the device set is a bit mask.  One engine + one host thread per device while a call runs; no exchange
 * between devices except the final reduction of the per-row statistics (pe_hip_sweep_reduce: combined on the host in device
 * order, bitwise reproducible).  Tables as pe_hip_load_circuit: batched parameter blocks are [batch][count][ncol] over the
 * WHOLE sweep.  All calls return pe_hip_status; pe_hip_sweep_last_error(s) is valid until the next call on s (s may be NULL
 * after a failed create). */
typedef struct pe_hip_sweep pe_hip_sweep;
int pe_hip_sweep_create(unsigned device_mask, pe_hip_sweep** out); /* bit d = HIP device d */
void pe_hip_sweep_destroy(pe_hip_sweep* s);
const char* pe_hip_sweep_last_error(pe_hip_sweep* s);
int pe_hip_sweep_devices(pe_hip_sweep* s);
int pe_hip_sweep_shard(pe_hip_sweep* s, int index, int* device, int* first_instance, int* count); /* block of the index-th device of the mask */
int pe_hip_sweep_set_options(pe_hip_sweep* s, const pe_hip_options* opt);
int pe_hip_sweep_load_circuit(pe_hip_sweep* s, int n_nodes, int n_branches, int batch, int n_tables, const pe_hip_device_table* tables);
int pe_hip_sweep_reset(pe_hip_sweep* s);
int pe_hip_sweep_operating_point(pe_hip_sweep* s, int mode, pe_hip_run_stats* stats); /* OP / DC / TROP solve of every instance */
int pe_hip_sweep_run(pe_hip_sweep* s, double dt, int nsteps, pe_hip_run_stats* stats); /* transient steps of every instance; stats: sums, slowest device's times */
int pe_hip_sweep_reduce(pe_hip_sweep* s, double* out); /* [4][rows]: sum, sum of squares, min, max of the current solution over all instances */
int pe_hip_sweep_get_solution(pe_hip_sweep* s, int first_instance, int count, double* x);
int pe_hip_sweep_get_instance_state(pe_hip_sweep* s, int first_instance, int count, int* status, long long* steps, long long* iters, double* t);
/* iteration count of every step of instance 0 since the last reset (parity with the reference's Newton counts) */
int pe_hip_get_newton_trace(pe_hip_engine* h, int capacity, int* iters, int* n_out);
/* last stamped MNA system of one instance (CSR, sorted columns; vals/rhs may be NULL) */
int pe_hip_get_matrix(pe_hip_engine* h, int instance, int* row_ptr, int* col_ind, double* vals, double* rhs);

/* in-kernel phase clocks of one instance since the last reset, 100 MHz ticks:
 * [0] device eval + MNA gather, [1] LU wave fronts (incl. the fused forward substitution), [2] LU cooperative fronts,
 * [3] cooperative part of the backward pass, [4] Newton bookkeeping, [5] backward pass (all of it; a separate forward pass of
 * the factor-reuse path is counted here too), [6] cooperative assembly, [7] cooperative block loops */
int pe_hip_get_phase_clocks(pe_hip_engine* h, int instance, long long* ticks8);
/* the same plus, from slot 8 on, six values per cooperative-front layout (0 whole front in LDS, 1 pivot panels + pulled Schur
 * tiles, 2 chain link): assembly, block loop, Schur update, factor store [ticks], fronts [count], sum of m*m; slots 48.. (split
 * schedule): for parts 0..3 of the instance, start / end tick and hardware placement of that workgroup in the last factor launch */
int pe_hip_get_phase_clocks_ex(pe_hip_engine* h, int instance, int capacity, long long* ticks, int* n_out);

/* host-only: run the symbolic analysis on a pattern and report its statistics (no GPU needed) */
int pe_hip_analyze_pattern(int n, const int* row_ptr, const int* col_ind, const double* values, pe_hip_info* out);

/* host-only: the assembly tree of that analysis.  Arrays of `capacity` ints; *n_fronts receives the count.
 * pivots[s], updates[s] (front order m = pivots + updates), parent[s] (-1 = root); fronts are in postorder. */
int pe_hip_analyze_pattern_fronts(int n, const int* row_ptr, const int* col_ind, const double* values, int capacity, int* pivots, int* updates,
                                  int* parent, int* n_fronts);

#ifdef __cplusplus
}
#endif
#endif /* PE_HIP_H */
