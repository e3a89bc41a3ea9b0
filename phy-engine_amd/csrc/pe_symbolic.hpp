// pe_symbolic.hpp -- host-side symbolic analysis for the MI355X multifrontal LU.
//
// Replaces the per-call `solver.compute(A)` analysis of the reference (Eigen SparseLU analyzePattern:
// COLAMD + etree + supernodes, circuits/circuit.h:1516) with a ONE-TIME analysis whose output is a static
// assembly tree of dense fronts.  The numeric work (pe_kernels.hip) then runs with no pivot search:
//   1. row matching   -> zero-free, large diagonal (MNA branch rows of V-sources have D == 0)
//   2. nested dissection (BFS level-set bisection) + minimum degree on the leaves, on pattern(PA + (PA)^T)
//   3. elimination tree, postorder, column structures, relaxed supernodes  -> fronts
//   4. assembly maps (A slot -> front cell, child update rows -> parent rows), storage offsets
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pe
{
    // Leading dimension of a column-major front image / panel in LDS: odd, so that the lanes of an access that walks ACROSS
    // columns (the column solves of a block step, the B operand of the MFMA tiles: stride = leading dimension) spread over the
    // banks instead of piling onto gcd(2 ld, 64) of them -- ld = 32 puts all 64 lanes of such an access on one bank pair.
    constexpr int pe_ld(int n) { return n | 1; }
    // LDS doubles the layout of a front occupies in front_factor (pe_front.hpp): the whole image (mode 0) or the pivot panels (modes 1, 2)
    // + the right-hand-side column of the fused forward substitution; a chain link continued in its child's image (mode 3) adds nothing
    constexpr long long front_lds_need(int mode, int p, int u)
    {
        long long const m = p + u;
        if(mode == 3) return 0;
        return (mode == 0 ? static_cast<long long>(pe_ld(p + u)) * m : static_cast<long long>(pe_ld(p + u)) * p + static_cast<long long>(pe_ld(p)) * u) + m;
    }

    struct SymbolicOptions
    {
        int nd_leaf{24};          // stop dissecting below this many vertices (large circuits: 10, pe_engine_policy.cpp symbolic_options, profiles/sweep_r02_leaf.log)
        int relax_small{8};       // always merge a last child into its parent while the merged front has <= this many pivots
        double relax_zero_frac{0.30};  // otherwise merge only if the explicit zeros added stay below this fraction of the merged panel
        int max_pivots{48};       // never grow a front beyond this many pivots (chains are split)
        double match_diag_rel{1e-8};   // keep a_jj as pivot when |a_jj| >= rel * max|row|
        // GPU mapping limits (see pe_front.hpp): a WAVE front (m <= wave_m, p <= wave_p) is factorised by one
        // wavefront in its own LDS slot; larger fronts are COOPERATIVE: the whole workgroup, pivot panels in LDS
        int wave_m{48};
        int wave_p{24};
        long long wave_slot{0};   // LDS doubles of one wavefront's slot; 0: (pe_ld(wave_m) + 1) * wave_m, i.e. every wave front whole.
                                  // Smaller: fronts up to wave_m whose image does not fit run in the panel layout, if their panels fit
        int absorb_m{48};         // a parent absorbs any child while the merged front order stays <= absorb_m (clamped to wave_m)
        int n_waves{8};           // wavefronts per workgroup (static assignment of wave subtrees)
        double cut_factor{1.0};   // a wave subtree may cost at most part_total / (cut_factor * n_waves)
        int n_parts{1};           // > 1: multi-workgroup mode, the tree below the level-1 cut is spread over this many workgroups
        double part_cut{1.5};     // a part subtree may cost at most total / (n_parts * part_cut)
        int shared_cu{};                 // the launch geometry keeps several workgroups per CU (128-VGPR kernel variants)
        int quad{};                      // 1: the wave fronts run on the lane-group kernel (pe_quad.hpp) where order <= 32, <= 16 pivots, <= 255 own entries of A, <= 16 children hold for a whole subtree;
                                         // the symbolic analysis then also builds the quad plan below and pads the arena with a zero region
        int quad_mid{};                  // 1: fronts of order 33..64 whose subtree qualifies form the MID class (f_kind 3) of a second lane-group launch
        int quad_lds_doubles{};          // LDS doubles per instance of a wavefront's update-matrix stack (0: every update matrix goes through the arena)
        // Second pass of the analysis (pe_engine_policy.cpp analyze_fitting): unknowns whose fronts sat at top levels that run ONE workgroup
        // per CU anyway (k_m2_factor_top_wide).  Those fronts own a CU's LDS: supernodes made of such unknowns only may take
        // max_pivots_top pivots and are split against panel_doubles_top, in links of equal length -- a separator of the top then is one
        // or two fronts instead of a chain of LDS-sized links, each of which costs a front's fixed work and a trip of its Schur block
        // through HBM.  Null: every front obeys max_pivots / panel_doubles.
        // Entry 1: a CU's whole LDS (panel_doubles_top); 2: half of it (panel_doubles_mid: levels that run two 8-wavefront workgroups
        // per CU, k_m2_factor_top_mid); a supernode takes the smallest share among its unknowns.
        std::vector<char> const* big_unknowns{nullptr};
        int max_pivots_top{64};
        long long panel_doubles_top{0};
        long long panel_doubles_mid{0};
    long long panel_reserve{384};    // LDS doubles kept free behind the panels of a large front (right-hand-side column, staged child maps)
    long long panel_doubles{18000};  // LDS doubles available for the L (m x p) and U (p x u) panels of a cooperative front
    };

    struct Symbolic
    {
        int n{};
        int nnzA{};
        // permutations (see pe_symbolic.cpp header): permuted row k takes original row row_src[k]; permuted column k is
        // original unknown col_src[k]
        std::vector<int> row_src, col_src;

        // fronts, in postorder (children before parents)
        int nfronts{};
        std::vector<int> f_col0;      // first pivot (permuted index)
        std::vector<int> f_p;         // pivots
        std::vector<int> f_u;         // update rows (m = p + u)
        std::vector<int> f_parent;    // -1 for roots
        std::vector<int> f_rows_ptr;  // [nfronts+1] into f_rows: the u update indices (permuted, ascending)
        std::vector<int> f_rows;
        std::vector<int> f_child_ptr, f_child;  // children lists (postorder)
        std::vector<int> f_rel_ptr, f_rel;      // for front s: local index in parent's front of each of its update rows (size u)
        std::vector<int> f_asm_ptr;             // [nfronts+1] into asm_slot/asm_pos
        std::vector<int> asm_slot;              // CSR slot of A
        std::vector<int> asm_pos;               // (r << 16) | c local cell
        std::vector<long long> f_lptr;          // offset of the m x p panel (column major, ld m) in the factor store
        std::vector<long long> f_uptr;          // offset of the p x u panel (column major, ld p)
        std::vector<long long> f_sptr;          // offset of the u x u update matrix (followed by the u update vector) in the stack arena
        long long factor_doubles{};
        long long arena_doubles{};
        long long work_doubles{};               // (unused)
        long long wave_panel_doubles{};         // largest p*(m+u) of a wave front (LDS doubles of one wavefront's slot)
        std::vector<int> f_kind;                // 0: wave front, 1: cooperative front of a part, 2: top front, 3: MID front (quad mode: a front of a
                                                // part's cooperative list of order <= 64 whose whole subtree the lane-group kernels can run; factored by
                                                // the second lane-group launch, the solve passes treat it as a cooperative front)
        std::vector<int> f_quad;                // 1: a wave front factored by the lane-group kernel (quad mode: order <= 32, <= 16 pivots, ... for its whole subtree)
        std::vector<int> f_wstack, f_wpar;      // wave fronts: offset of their solved vector on the wavefront's backward stack, and their parent's (-1: subtree root)
        int wave_stack{1};                      // doubles of the deepest stack (sum of the front orders along a path inside a wave subtree)
        int n_parts{1};
        // part q, wavefront w: wave_list[wave_ptr[q*(W+1)+w] .. wave_ptr[q*(W+1)+w+1]) in postorder (entry W of a part is empty)
        std::vector<int> wave_ptr, wave_list;
        std::vector<int> coop_ptr, coop_list;   // cooperative fronts of part q: coop_list[coop_ptr[q] .. coop_ptr[q+1])
        std::vector<int> top_ptr, top_list;     // top fronts of level l (0-based): top_list[top_ptr[l] .. top_ptr[l+1]); children before parents
        // pull-based assembly of the Schur blocks: for child edge e (position in f_child),
        // f_inv[f_inv_off[e] + r] = index of parent-local row r among the child's update rows, or -1
        std::vector<long long> f_inv_off;
        std::vector<int> f_cnp;                 // per child edge: how many of the child's update rows are pivots of the parent (a prefix: f_rel ascends)
        std::vector<int> f_inv;
        std::vector<unsigned> f_bmask;          // per child edge: bit min(t, 31) set when the child has an update row among the parent's update rows 16 t .. 16 t + 15
        int max_m{};                            // largest front order
        int max_u{};

        // Destination-centric assembly lists (build_assembly_lists, after the launch geometry fixed the LDS caps).  For a
        // front s in whole-front or panel layout every LDS cell that receives a contribution from a child's update matrix
        // (or update vector: the right-hand-side column behind the panels) is ONE cell of the list:
        //   gl_dst[gl_ptr[s] + c]                      its offset in the front's LDS image,
        //   gl_src[gl_sptr[s] + (n_0 + .. + n_{r-1}) + c]  the arena offset of its r-th source (children in list order), c < n_r,
        //   n_r = gl_cnt[gl_rptr[s] + r]               cells with more than r sources (cells are sorted by source count, descending,
        //                                              then by the address of the first source: coalesced loads of the first child).
        // One sweep without barriers assembles all children; the summation order of a cell is the children's order.
        std::vector<int> f_mode;                // 0 whole front in LDS, 1 pivot panels in LDS + pulled Schur tiles, 2 chain link,
                                                // 3 chain link whose front is already in LDS: the Schur block its child left there (f_keep)
        std::vector<int> f_keep;                // per front: 1 = the parent is a mode-3 front (the Schur block is not written to the arena)
        std::vector<int> gl_ptr, gl_rptr;       // [nfronts + 1]
        std::vector<long long> gl_sptr;         // [nfronts + 1]
        std::vector<unsigned short> gl_dst;
        std::vector<int> gl_cnt, gl_src;

        // ---- quad plan (opt.quad): the index "program" of the lane-group kernel for the wave fronts (pe_quad.hpp).  A wavefront handles
        // FOUR instances (16 lanes each: lane = 16 q + r), the front in registers (row r + 16 s of row set s, every column), all index
        // data wavefront-uniform.  Wave-front list L (= part * n_waves + wavefront) is a run of blocks in q_prog, fronts in list order:
        //   q_lists[2 L] first int of the run, q_lists[2 L + 1] fronts in it.
        // Block of a front = Q_HDR ints, then Q_CHILD ints per child:
        //   hdr  [0] m  [1] p  [2] first pivot  [3] first own entry of A (f_asm_ptr)  [4] children  [5] row sets RS (1: m <= 16, 2: m <= 32)
        //        [6,7] f_lptr lo / hi  [8,9] f_sptr lo / hi  [10] byte offset of its per-lane data in q_lane
        //        [11] RS of the NEXT front of the list (0: none)  [12] byte offset of that front's per-lane data (prefetched one front ahead)
        //        [13] where its update matrix + vector go: offset (doubles) on the wavefront's LDS stack, or -1: the arena slot f_sptr
        //   child [0] its f_sptr (doubles, < 2^31)  [1] its u  [18] its LDS stack offset or -1 (= its own hdr[13])  [2..17] 64 bytes (32 used by the wave fronts): [C] = 1 + column of the child's update matrix that lands in
        //        column C of this front, 0 = none (rows and columns share the map: structurally symmetric fronts)
        // q_lane, per front 16 RS row records of 16 RS + 16 bytes: [C] (C < 16 RS) = 1 + index (relative to hdr[3]) of the entry of A in
        // cell (row, C), 0 = none; then [16 RS + e] = 1 + row of child e's update matrix that lands in this row, 0 = none (e < 16).
        enum : int
        {
            Q_HDR = 16,
            Q_CHILD = 32,   // (child: [2..17] = 64 column bytes for the mid fronts, [18] its LDS stack offset)
            Q_ZERO = 4224   // doubles of the zero region behind every instance's arena: where lanes without a contribution read (64 x 64 + 64 + slack)
        };
        // LDS stack: an update matrix whose parent sits in the same list stays on chip when it fits -- lists are walked in postorder, so
        // the children of a front are always the most recent unconsumed entries (LIFO); q_lds_doubles = doubles of one instance's stack.
        int quad{};
        int q_lds_doubles{};
        long long q_lds_kept{}, q_lds_total{};  // doubles of update matrices (+ vectors) of wave fronts kept in LDS / in all (statistics)
        std::vector<int> q_prog, q_lists;
        std::vector<unsigned char> q_lane;
        // backward pass of the quad fronts (k_m2_backward_quads): per list the same fronts in REVERSE order (parents first), Q_BACK ints each:
        //   [0] m  [1] p  [2] first pivot  [3] u  [4,5] f_lptr lo / hi  [8 .. 8 + u) permuted indices of the update rows (f_rows)
        enum : int
        {
            Q_BACK = 40
        };
        std::vector<int> q_bprog;
        // the same for the MID fronts (f_kind 3; row sets 1..4): q2_lists[2 L] / [2 L + 1] for list L = part * n_waves + k
        std::vector<int> q2_prog, q2_lists;
        std::vector<unsigned char> q2_lane;
        int n_mid{};
        long long q_zero_off{};                 // offset of the zero region in the arena (doubles)

        // statistics
        long long nnz_LU{};      // structural nnz(L)+nnz(U) (diagonal counted once) of the supernodal pattern WITHOUT relaxation zeros
        long long nnz_LU_stored{};  // with relaxation zeros (what the dense panels hold)
        double flops{};          // dense front flops (2*mul-add) of one numeric factorisation
        int tree_depth{};
        int n_row_swaps{};       // rows whose pivot is not their own diagonal
        bool structurally_singular{};
        std::string error;
    };

    // rp/ci: CSR pattern with sorted columns; vals may be null (then every entry weighs 1).
    bool analyze(int n, int const* rp, int const* ci, double const* vals, SymbolicOptions const& opt, Symbolic& out);

    // cap_wave / cap_team: LDS doubles of a wavefront's slot / of the whole workgroup (the `cap` front_factor is called with).
    // Returns false (S.error set) when an offset does not fit its index type.
    // `top_wide` (may be null): per top LEVEL, 1 = one workgroup per front with `cap_top` doubles of LDS (the 16-wavefront launch),
    // 2 = the same and every front of the level laid out against cap_top, 3 = the 8-wavefront launch with `cap_mid` doubles
    bool build_assembly_lists(Symbolic& S, long long cap_wave, long long cap_team, int const* top_wide = nullptr, long long cap_top = 0, long long cap_mid = 0);
}  // namespace pe
