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
    struct SymbolicOptions
    {
        int nd_leaf{48};          // stop dissecting below this many vertices
        int relax_small{8};       // always merge a last child into its parent while the merged front has <= this many pivots
        double relax_zero_frac{0.30};  // otherwise merge only if the explicit zeros added stay below this fraction of the merged panel
        int max_pivots{96};       // never grow a front beyond this many pivots (chains are split)
        double match_diag_rel{1e-8};   // keep a_jj as pivot when |a_jj| >= rel * max|row|
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
        std::vector<long long> f_sptr;          // offset of the u x u update matrix in the stack arena
        long long factor_doubles{};
        long long arena_doubles{};
        int max_m{};                            // largest front order
        int max_u{};

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
}  // namespace pe
