// pe_engine_seam.cpp -- the complex twin of the solver seam: pe_hip_solve_csr_complex, the drop-in for
// cuda_sparse_lu::solve_csr_timed / solve_csr on std::complex<double> (reference: circuits/solver/cuda_sparse_lu.h:295-312), which
// circult::solve_once calls whenever the stamped system is not all-real (circuit.h:1332: AC / ACOP).
//
// The complex system (Ar + j Ai)(xr + j xi) = br + j bi of n unknowns is solved in real-equivalent form
//        [ Ar  -Ai ] [ xr ]   [ br ]
//        [ Ai   Ar ] [ xi ] = [ bi ]          (2n unknowns, 4 nnz entries)
// by the kernels of the real seam (symbolic analysis once per pattern, k_factor_solve), followed by fp64 iterative refinement on the
// device exactly as pe_hip_analyze_ac does it (pe_engine_ac.cpp): the static pivot order separates the two halves of a complex pivot,
// refinement brings the componentwise backward error back to rounding level.  A cached pattern (copy_pattern == 0) keeps the pivot
// order of the values it was analysed on; a bad pivot or a refinement that does not converge re-analyses on the current values once.
#include "pe_engine_internal.hpp"

using namespace pe_eng;

namespace
{
    // real-equivalent CSR pattern of a complex CSR pattern with sorted columns: row i = [cols of row i | the same + n], row n + i likewise;
    // entry k of the complex matrix lands at the four positions pos[4k..4k+3] = (Ar in row i, -Ai in row i, Ai in row n+i, Ar in row n+i)
    void real_equivalent_pattern(int n, int const* rp, int const* ci, std::vector<int>& rp2, std::vector<int>& ci2, std::vector<int>& pos)
    {
        int const nnz = rp[n];
        rp2.assign(2 * static_cast<size_t>(n) + 1, 0);
        ci2.resize(4 * static_cast<size_t>(nnz));
        pos.resize(4 * static_cast<size_t>(nnz));
        for(int i = 0; i < n; ++i)
        {
            int const len = rp[i + 1] - rp[i];
            rp2[i + 1] = 2 * len;
            rp2[n + i + 1] = 2 * len;
        }
        for(int r = 0; r < 2 * n; ++r) rp2[r + 1] += rp2[r];
        for(int i = 0; i < n; ++i)
        {
            int const len = rp[i + 1] - rp[i], top = rp2[i], bot = rp2[n + i];
            for(int k = 0; k < len; ++k)
            {
                int const e = rp[i] + k, c = ci[e];
                ci2[top + k] = c;
                ci2[top + len + k] = c + n;
                ci2[bot + k] = c;
                ci2[bot + len + k] = c + n;
                pos[4 * static_cast<size_t>(e) + 0] = top + k;
                pos[4 * static_cast<size_t>(e) + 1] = top + len + k;
                pos[4 * static_cast<size_t>(e) + 2] = bot + k;
                pos[4 * static_cast<size_t>(e) + 3] = bot + len + k;
            }
        }
    }
}  // namespace

extern "C" {

int pe_hip_solve_csr_complex(pe_hip_engine* h, int n, int nnz, const int* row_ptr, const int* col_ind, const double* values_re_im, const double* b_re_im,
                             double* x_re_im, int copy_pattern, pe_hip_timings* out)
{
    if(!h || n < 0 || nnz < 0 || !row_ptr || !col_ind || !values_re_im || !b_re_im || !x_re_im) return PE_HIP_ERR_ARG;
    if(n > 0 && (row_ptr[0] != 0 || row_ptr[n] != nnz)) return fail(h, PE_HIP_ERR_ARG, "solve_csr_complex: row_ptr[0] != 0 or row_ptr[n] != nnz");
    if(static_cast<long long>(nnz) * 4 > 0x7fffffffll || static_cast<long long>(n) * 2 > 0x7fffffffll)
        return fail(h, PE_HIP_ERR_ARG, "solve_csr_complex: the real-equivalent system exceeds int32 indices");
    auto const t_total = clk::now();
    pe_hip_timings tm{};
    HIPCHK(h, hipSetDevice(h->device));
    if(n == 0) return PE_HIP_OK;
    auto& C = h->csrz;
    int const n2 = 2 * n, nnz2 = 4 * nnz;
    // the values of this call in real-equivalent CSR order
    bool const fresh = copy_pattern || !C.have || C.n != n || C.nnz != nnz;
    if(fresh)
    {
        if(char const* bad = csr_pattern_error(n, nnz, row_ptr, col_ind)) return fail(h, PE_HIP_ERR_ARG, std::string("solve_csr_complex: ") + bad);
        real_equivalent_pattern(n, row_ptr, col_ind, C.rp2, C.ci2, C.pos);
    }
    C.vals.resize(static_cast<size_t>(nnz2));
    for(int e = 0; e < nnz; ++e)
    {
        double const re = values_re_im[2 * static_cast<size_t>(e)], im = values_re_im[2 * static_cast<size_t>(e) + 1];
        int const* q = &C.pos[4 * static_cast<size_t>(e)];
        C.vals[q[0]] = re;
        C.vals[q[1]] = -im;
        C.vals[q[2]] = im;
        C.vals[q[3]] = re;
    }
    C.rhs.resize(static_cast<size_t>(n2));
    for(int i = 0; i < n; ++i)
    {
        C.rhs[i] = b_re_im[2 * static_cast<size_t>(i)];
        C.rhs[n + i] = b_re_im[2 * static_cast<size_t>(i) + 1];
    }
    // symbolic analysis (static pivot order matched on |values| of THIS call) + device tables
    auto analyse = [&]() -> int
    {
        auto const t0 = clk::now();
        C.have = false;
        C.pool.release();
        pe::SymbolicOptions so{};
        if(int const rc = analyze_fitting(h, 1, 0, n2, C.rp2.data(), C.ci2.data(), C.vals.data(), C.sym, so); rc != PE_HIP_OK) return rc;
        pe::DevView V{};
        V.rows = n2;
        V.n_nodes = n2;
        V.batch = 1;
        V.nnzA = nnz2;
        if(int const rc = upload_symbolic(h, C.pool, C.sym, so, V, 1); rc != PE_HIP_OK) return rc;
        HIPCHK(h, C.pool.alloc(V.aval, static_cast<size_t>(nnz2)));
        HIPCHK(h, C.pool.alloc(V.rhs, static_cast<size_t>(n2)));
        HIPCHK(h, C.pool.alloc(V.x, static_cast<size_t>(n2)));
        HIPCHK(h, C.pool.alloc(V.w, static_cast<size_t>(n2)));
        HIPCHK(h, C.pool.alloc(V.status, 1));
        HIPCHK(h, C.pool.upload(V.csr_rp, C.rp2));  // refinement residual: A row by row in CSR order (aval is kept in that order here)
        HIPCHK(h, C.pool.upload(V.csr_ci, C.ci2));
        V.slot_e = nullptr;
        HIPCHK(h, C.pool.alloc(C.d_xacc, static_cast<size_t>(n2)));
        HIPCHK(h, C.pool.alloc(C.d_b0, static_cast<size_t>(n2)));
        HIPCHK(h, C.pool.alloc(C.d_worst, 1));
        C.V = V;
        C.n = n;
        C.nnz = nnz;
        C.have = true;
        C.on_these_values = true;
        tm.analyze_ms += ms_since(t0);
        return PE_HIP_OK;
    };
    if(fresh)
    {
        if(int const rc = analyse(); rc != PE_HIP_OK) return rc;
    }
    else
        C.on_these_values = false;
    float gpu_ms = 0.f;
    // one factor + solve of the current V.rhs; returns the device status word (0 ok) in `status`
    auto factor_solve = [&](int& status) -> int
    {
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        HIPCHK(h, pe::launch_factor_solve(h->stream, C.V, true));
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipMemcpyAsync(&status, C.V.status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        gpu_ms += ms;
        return PE_HIP_OK;
    };
    // solve + refine on the device: x in C.d_xacc, `worst` = componentwise backward error reached; status != 0: bad pivot
    auto solve_refined = [&](int& status, double& worst) -> int
    {
        auto t0 = clk::now();
        HIPCHK(h, hipMemcpyAsync(C.V.aval, C.vals.data(), static_cast<size_t>(nnz2) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(C.V.rhs, C.rhs.data(), static_cast<size_t>(n2) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        tm.h2d_ms += ms_since(t0);
        t0 = clk::now();
        worst = 0.0;
        if(int const rc = factor_solve(status); rc != PE_HIP_OK) return rc;
        if(status == 0)
        {
            HIPCHK(h, pe::launch_ac_accumulate(h->stream, C.V, C.d_xacc, C.d_b0, true));
            for(int round = 0;; ++round)
            {
                HIPCHK(h, pe::launch_csr_residual(h->stream, C.V, C.d_xacc, C.d_b0, C.d_worst));
                HIPCHK(h, hipMemcpyAsync(&worst, C.d_worst, sizeof(double), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if(!(worst > 4.0e-16) || round == 3) break;  // (NaN: falls through to the rounds, ends above tolerance)
                if(int const rc = factor_solve(status); rc != PE_HIP_OK) return rc;
                if(status != 0) break;
                HIPCHK(h, pe::launch_ac_accumulate(h->stream, C.V, C.d_xacc, C.d_b0, false));
            }
        }
        tm.solve_host_ms += ms_since(t0);
        return PE_HIP_OK;
    };
    constexpr double accept = 1.0e-10;  // componentwise backward error a returned solution may carry (healthy solves end at ~1e-16)
    int status = 0;
    double worst = 0.0;
    if(int const rc = solve_refined(status, worst); rc != PE_HIP_OK) return rc;
    if((status != 0 || !(worst <= accept)) && !C.on_these_values)
    {
        // the cached pivot order was matched on another call's values: once more on these
        if(int const rc = analyse(); rc != PE_HIP_OK) return rc;
        if(int const rc = solve_refined(status, worst); rc != PE_HIP_OK) return rc;
    }
    tm.solve_ms = gpu_ms;
    if(status != 0) return fail(h, PE_HIP_ERR_SINGULAR, "solve_csr_complex: singular matrix (zero / non-finite pivot)");
    if(!(worst <= accept)) return fail(h, PE_HIP_ERR_INACCURATE, "solve_csr_complex: backward error " + std::to_string(worst) + " after refinement");
    auto const t0 = clk::now();
    C.x.resize(static_cast<size_t>(n2));
    HIPCHK(h, hipMemcpy(C.x.data(), C.d_xacc, static_cast<size_t>(n2) * sizeof(double), hipMemcpyDeviceToHost));
    for(int i = 0; i < n; ++i)
    {
        x_re_im[2 * static_cast<size_t>(i)] = C.x[i];
        x_re_im[2 * static_cast<size_t>(i) + 1] = C.x[n + i];
    }
    tm.d2h_ms = ms_since(t0);
    tm.total_host_ms = ms_since(t_total);
    if(out) *out = tm;
    return PE_HIP_OK;
}

}  // extern "C"
