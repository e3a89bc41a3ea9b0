// pe_device.hpp -- plain-data views shared by the host engine and the HIP kernels.
#pragma once
#include <cstdint>

#include "pe_symbolic.hpp"  // pe_ld

namespace pe
{
    // dv ("device values") is the per-instance vector every matrix / RHS contribution is gathered from.
    // Fixed slots first, then one block per device kind (offsets in DevView).
    enum : int
    {
        DV_ONE = 0,   // 1.0  (incidence entries of B / C)
        DV_GMIN = 1,  // env.g_min (circuit.h:1107-1110)
        DV_FIXED = 2
    };
    // in-kernel phase clocks per instance: [0] eval+stamp, [1] LU wave fronts, [2] LU cooperative fronts, [3] forward, [4] -,
    // [5] backward, [6] cooperative assembly, [7] cooperative block loop; then per cooperative front layout L = 0 whole-front,
    // 1 panel+pull, 2 chain link at 8 + 6 L: assembly, block loop, Schur, store, fronts, sum of m*m; [26..31] wave-phase time of
    // wavefronts 0..5 of part 0; [32..38] the same six figures (+ own-entries share of the assembly) for the wave fronts of wavefront 0 of part 0; [40..47] factorisation time of parts 0..7
    enum : int
    {
        PE_PROF = 96  // [64..79] lane-group kernel (pe_quad.hpp), list 0 of the quad that starts at this instance: header wait, assembly, elimination, stores, whole fronts, fronts (count)
    };

    // diode parameter columns after host-side prepare_foundation (PN_junction.h:296-354)
    enum : int
    {
        DP_IS_EFF = 0,
        DP_ISR_EFF,
        DP_UTE,     // N * Ut
        DP_UTER,    // Nr * Ut
        DP_UTH,
        DP_BV_EFF,
        DP_BV_SET,  // 0 / 1
        DP_TT,      // transit time used by step_changed_tr
        DP_TT_STAMP,  // 1: iterate_tr stamps the diffusion-cap companion, 0: DC stamp only (FBR)
        DP_NCOL
    };

    struct DevView
    {
        // ---- sizes
        int rows, n_nodes, n_branches, batch;
        int nnzA, dv_len;
        int nR, nC, nL, nVdc, nVac, nIdc, nD, nDrv;
        int nonlinear;
        int nTs, nCl, nN3, nRl;  // time sources (IAC, generators), coupled-inductor pairs, three-pin non-linear devices (MOSFET, BJT), relays
        // ---- dv block offsets
        int dv_r, dv_cg, dv_ci, dv_lr, dv_lu, dv_vdc, dv_vac, dv_idc, dv_dg, dv_di, dv_drv;
        // ---- topology (shared by all instances); rows are MNA row indices, -1 = ground
        int const *c_a, *c_b;
        int const *l_a, *l_b, *l_k;  // l_k = absolute row of the branch
        int const *vac_k;
        int const *d_a, *d_c;
        int const *ts_kind, *ts_dv;   // time source i: 0 = IAC, 1.. = generator type + 1; dv slot of its value
        int const *cl_n, *cl_k;       // coupled inductors i: rows of p1,p2,s1,s2 [.][4]; absolute rows of the two branches [.][2]
        int const* cl_dv;             // first of its five dv slots: req11, req12, req22, Ueq1, Ueq2
        int const *n3_kind, *n3_n, *n3_dv;  // MOSFET / BJT i: pe_hip_kind; rows of its pins [.][3]; first dv slot (MOS: gds, gm, Ieq; BJT: geq, Ieq_be, gm, Ieq_c)
        // ---- stamping: CSR of contributions per A slot / per RHS row; entry = (dv index << 1) | negate
        int const *a_ptr, *a_src;
        int const *b_ptr, *b_src;
        // slots of aval / rows of rhs with a contribution that depends on x (junctions, MOS / BJT, relays, host-stamp overlay): the
        // only ones that change between the Newton iterations of one solve point (null: not built -- always stamp everything)
        int const *dyn_a{}, *dyn_b{};
        int n_dyn_a{}, n_dyn_b{};
        // ---- per-instance parameters  [batch][count(*cols)]
        double const* c_cap;
        double const* l_ind;
        double const* vac_par;  // [.][nVac][3] Vp, omega, phase
        double const* d_par;    // [.][nD][DP_NCOL]
        double const* ts_par;   // [.][nTs][8]  IAC: Ip, omega, phase; generator: type, Vh, Vl, freq, duty, phase, tr, tf
        double const* cl_par;   // [.][nCl][3]  L1, L2, k
        double const* n3_par;   // [.][nN3][3]  MOS: Kp, lambda, Vth; BJT: Is*Area, N*Ut, BetaF
        int const *rl_n, *rl_dv; // relay i: rows of its coil pins [.][2]; dv slot of its contact resistance
        double const* rl_par;   // [.][nRl][2]  Von, Voff
        int* rl_engaged;        // [.][nRl]     contact state (relay.h:85-92)
        double r_open;          // contact resistance of an open relay / switch
        // ---- per-instance state
        double *c_hist, *c_prevg;          // [.][nC]
        double *d_udlast, *d_geq, *d_hist, *d_prevg;  // [.][nD]
        double* dv;     // [.][dv_len]
        double* aval;   // [.][nnzA]
        double* rhs;    // [.][rows]
        double* x;      // [.][rows]   node voltages ; branch currents
        double* xprev;  // [.][rows]   previous Newton iterate
        double* w;      // [.][rows]   permuted work vector of the triangular solves
        double* factor; // [.][factor_doubles]
        double* arena;  // [.][arena_doubles]  update-matrix stack
        double* t_now;     // [.]
        double* last_step; // [.]
        int* status;       // [.]  pe_hip_status
        long long* n_steps;  // [.]
        long long* n_iters;  // [.]
        long long* prof;     // [.][8] phase clocks (100 MHz ticks): eval+stamp, factor waves, factor coop, fwd waves, fwd+bwd coop, bwd waves, newton, companion
        int* trace;          // Newton iterations per step of instance 0
        int trace_cap;
        int* trace_len;
        long long factor_doubles, arena_doubles;
        // ---- symbolic (shared)
        int nfronts;
        int const *f_col0, *f_p, *f_u;
        int const *f_rows_ptr, *f_rows;
        int const *f_child_ptr, *f_child;
        int const* f_rel;       // indexed through f_rows_ptr
        long long const* f_inv_off;  // per child edge (index into f_child), cooperative parents only
        int const* f_cnp;            // per child edge: leading update rows of the child that are pivot rows of the parent
        int const* f_inv;
        unsigned const* f_bmask;     // per child edge: 16-row blocks of the parent's update rows the child touches (bit min(t, 31))
        int const *f_asm_ptr, *asm_slot, *asm_pos;
        // destination-centric assembly lists (pe_symbolic.hpp: build_assembly_lists) and the LDS layout of every front
        int const* f_mode;             // 0 whole front, 1 pivot panels, 2 chain link, 3 chain link continued in LDS (top run, see f_keep)
        int const* f_keep{};           // per front: 1 = its Schur block stays in the LDS image for its parent (a mode-3 front, next in the same top run)
        int const *gl_ptr, *gl_rptr;   // [nfronts + 1] into gl_dst / gl_cnt
        long long const* gl_sptr;      // [nfronts + 1] into gl_src
        unsigned short const* gl_dst;
        int const *gl_cnt, *gl_src;
        double* zero;  // one 0.0 in device memory: where the masked-out lanes of an unconditional gather point
        long long const *f_lptr, *f_uptr, *f_sptr;
        int const *row_src, *col_src;
        // split schedule, round 4: the stamp kernel also initialises the permuted work vector (w[k] = rhs[row_src[k]], what k_m2_winit did in
        // a launch of its own): row_dst = inverse of row_src; row_dyn[r] != 0: row r is re-gathered by the x-dependent-only stamp (dyn_b)
        int const* row_dst{};
        unsigned char const* row_dyn{};
        // schedule (pe_symbolic.cpp): phase 1 = per-wavefront lists of small fronts, phase 2 = cooperative fronts
        // part q, wavefront w: wave_list[wave_ptr[q*(n_waves+1)+w] ..); cooperative fronts of part q: coop_list[coop_ptr[q] ..);
        // top fronts of level l: top_list[top_ptr[l] ..)  (multi-workgroup mode only)
        int const *wave_ptr, *wave_list, *coop_ptr, *coop_list, *top_ptr, *top_list;
        int n_parts, n_top_levels, n_waves;
        int top_cnt[64];        // fronts per top level (host-side copy for the launch geometry)
        int top_wide[64]{};     // 1 / 2: the level runs one 16-wavefront workgroup per front with a CU's whole LDS (k_m2_factor_top_wide; 2: it holds
                                // fronts formed against that LDS); 3: one 8-wavefront workgroup per front with half a CU's LDS (k_m2_factor_top_mid)
        int lds_top_doubles{};  // dynamic LDS of the 16-wavefront launches, in doubles (>= lds_doubles)
        int lds_mid_doubles{};  // ... of the 8-wavefront launches of the levels marked 3
        // launch-shape knobs of THIS engine (pe_hip_set_knob / PHY_ENGINE_HIP_*; read once per analysis into the view, never from the
        // environment at launch time): levels of at most mid_top_limit workgroups run the 8-wavefront launch; ew_grid > 0 forces the
        // workgroups per instance of the elementwise kernels; quad_lds_pad pads the lane-group kernel's LDS request (occupancy probe)
        int mid_top_limit{512}, ew_grid{}, quad_lds_pad{};
        int top_run_any_class{};  // TEST knob (TEST_OLD_TOP_RUNS): runs of single-front levels join levels of different LDS classes -- the
                                  // round-3 grouping bug, kept reachable so that the LDS guard can be shown to refuse it
        int const* f_need{};      // per front: doubles of LDS its layout occupies (image or panels + right-hand-side column; 0 for a chain
                                  // link continued in its child's image) -- compared with the LDS its launch really has (flag bit 3)
        int* active;            // [.] multi-workgroup mode: instances still iterating
        int* flags;             // [.] multi-workgroup mode: bit 0 non-finite solution, bit 1 Newton violation, bit 2 bad pivot,
                                //     bit 3 a front's LDS layout does not fit the launch that runs it (internal error, never "singular")
        int high_occupancy;     // 1: launch the 128-VGPR kernel variant (several workgroups per CU)
        int keep_l21;           // 1: a later launch may reuse the factors with a separate forward pass (linear circuit, refactor_every_solve = 0)
        int wave_m, wave_p, max_m, max_p;
        int lds_slot;           // doubles of one wavefront's panel slot (largest p*(m+u) of a wave front)
        int lds_sslot;          // doubles of one wavefront's solve scratch (+ its backward stack, lds_bstack_off doubles in)
        int lds_bstack_off{};   // the backward pass keeps the solved vectors of a wave front's ancestors inside the subtree on a stack:
        int const *f_wstack{}, *f_wpar{};  //   offsets of a wave front's vector and of its parent's (-1: root of the subtree)
        int lds_wave_stage{}, lds_coop_stage{};  // doubles of the staged block of a wavefront / of the workgroup in the triangular solves
        int lds_top_stage{};                     // ... of the workgroup in the triangular solves of the TOP fronts (>= lds_coop_stage: their pivot limit may be larger)
        int lds_solve_top_doubles{};             // dynamic LDS of k_m2_solve_top, in doubles
        int lds_doubles;        // dynamic LDS size of a launch, in doubles
        int lds_solve_doubles;  // ... of the triangular-solve kernels of the split schedule
        // lean plan of the backward pass (front_backward_lean: only U11 staged): slot, stack offset, staged block, launch size
        int lds_sslot_b{}, lds_bstack_off_b{}, lds_wave_stage_b{}, lds_solve_b_doubles{};
        // ---- lane-group kernel of the wave fronts (pe_quad.hpp; tables: pe_symbolic.hpp "quad plan")
        int quad{};                    // 1: the wave fronts of the split schedule run on k_m2_factor_quads, factor_part skips them
        int const *q_prog{}, *q_lists{};
        unsigned char const* q_lane{};
        int const* q_bprog{};               // backward pass of the quad fronts (pe_symbolic.hpp Q_BACK), lists reversed; 1 in quad_back: in use
        int quad_back{};
        int const *q2_prog{}, *q2_lists{};  // the MID fronts (f_kind 3): second lane-group launch
        unsigned char const* q2_lane{};
        int n_mid{};                        // MID fronts per instance (0: no second launch)
        int const* f_quad{};                // per front: 1 = a wave front the lane-group kernel factors (the per-instance wave phase skips it)
        int const* f_kind{};                // per front (pe_symbolic.hpp): factor_part skips the MID fronts of its cooperative list
        long long q_zero_off{};        // zero region of every instance's arena (doubles)
        int q_lds_stride{};            // doubles between the LDS stacks of two instances of a quad (0: no LDS stack); a launch needs 4 x 8 x this bytes
        int const* q_list{};           // [n_quads][4] instances of a quad (-1: none), ascending; all within one 32-bit byte-offset window of [0]
        int n_quads{};
        // ---- Newton
        double v_abstol, v_reltol, i_abstol, i_reltol;
        int max_newton;
        // ---- residual safety net of the static-pivot LU (pe_front.hpp residual_norms): CSR of A in ORIGINAL row / column order
        // (shared), slot_e[slot] = index of that entry in aval (front-assembly order; null = identity), per-instance buffers
        int const *csr_rp, *csr_ci, *slot_e;
        double* xsave;    // [.][rows]  solution being refined
        double* rres;     // [.][rows]  residual b - A x (right-hand side of the correction solve)
        double* eta_acc;  // [.][4]     max |r_i|, max_i sum_j |a_ij|, max |x_i|, max |b_i| of the last solve (split schedule: atomic max)
        double residual_tol;  // <= 0: check disabled
    };
}  // namespace pe
