// tests/cpp/hip_sparse_lu_seam.cpp -- the binding of INTEGRATION.md section A, compiled as written there, driven through the three
// members circult::solve_once uses of its solver (reference: circuit.h:1134 is_available, :1320 solve_csr_real, :1332 the complex
// solve_csr_timed; declared cuda_sparse_lu.h:295-312, 465-473).  Own systems with answers known by construction: x is chosen, b = A x.
// exit 0 = pass.
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include <pe_hip.h>

namespace phy_engine::solver
{
    struct hip_sparse_lu
    {
        struct timings { double h2d_ms{}, solve_ms{}, d2h_ms{}, solve_host_ms{}, solve_total_host_ms{}; };   // cuda_sparse_lu.h:27-34
        ::pe_hip_engine* h{};
        hip_sparse_lu() noexcept { (void)::pe_hip_create(0, &h); }
        hip_sparse_lu(hip_sparse_lu const&) = delete;
        hip_sparse_lu& operator=(hip_sparse_lu const&) = delete;
        ~hip_sparse_lu() { ::pe_hip_destroy(h); }
        [[nodiscard]] bool is_available() const noexcept { return h != nullptr; }                             // circuit.h:1134
        // identical signature to cuda_sparse_lu::solve_csr_real (cuda_sparse_lu.h:465-473), called at circuit.h:1320
        [[nodiscard]] bool solve_csr_real(int n, int nnz, int const* row_ptr, int const* col_ind, double const* values,
                                          double const* b, double* x, timings& out, bool copy_pattern = true) noexcept
        {
            ::pe_hip_timings t{};
            bool const ok = h && ::pe_hip_solve_csr_real(h, n, nnz, row_ptr, col_ind, values, b, x, copy_pattern ? 1 : 0, &t) == PE_HIP_OK;
            out = {t.h2d_ms, t.solve_ms, t.d2h_ms, t.solve_host_ms, t.total_host_ms};
            return ok;
        }
        // identical signature to the complex twin cuda_sparse_lu::solve_csr_timed (cuda_sparse_lu.h:304-312), called at circuit.h:1332;
        // std::complex<double> is layout-compatible with double[2] (the reference relies on the same fact: cuda_sparse_lu.h:314)
        [[nodiscard]] bool solve_csr_timed(int n, int nnz, int const* row_ptr, int const* col_ind, ::std::complex<double> const* values,
                                           ::std::complex<double> const* b, ::std::complex<double>* x, timings& out, bool copy_pattern = true) noexcept
        {
            ::pe_hip_timings t{};
            bool const ok = h && ::pe_hip_solve_csr_complex(h, n, nnz, row_ptr, col_ind, reinterpret_cast<double const*>(values),
                                                            reinterpret_cast<double const*>(b), reinterpret_cast<double*>(x), copy_pattern ? 1 : 0, &t) == PE_HIP_OK;
            out = {t.h2d_ms, t.solve_ms, t.d2h_ms, t.solve_host_ms, t.total_host_ms};
            return ok;
        }
        [[nodiscard]] bool solve_csr(int n, int nnz, int const* row_ptr, int const* col_ind, ::std::complex<double> const* values,
                                     ::std::complex<double> const* b, ::std::complex<double>* x, bool copy_pattern = true) noexcept   // cuda_sparse_lu.h:295-302
        { timings ignored{}; return solve_csr_timed(n, nnz, row_ptr, col_ind, values, b, x, ignored, copy_pattern); }
    };
}  // namespace phy_engine::solver

using cplx = std::complex<double>;

// a ladder of n nodes: series admittance y_s between neighbours, shunt y_p to ground at every node, a unit branch row at the end (an
// ideal source: zero diagonal, the entry the static pivot order has to match away) -- the shape an MNA system has
template <class T>
static void ladder(int n, T ys, T yp, std::vector<int>& rp, std::vector<int>& ci, std::vector<T>& va)
{
    int const N = n + 1;
    rp.assign(1, 0);
    ci.clear();
    va.clear();
    for(int i = 0; i < N; ++i)
    {
        if(i < n)
        {
            if(i > 0) { ci.push_back(i - 1); va.push_back(-ys); }
            T d = yp + (i > 0 ? ys : T{}) + (i + 1 < n ? ys : T{});
            ci.push_back(i); va.push_back(d * (1.0 + 0.01 * i));
            if(i + 1 < n) { ci.push_back(i + 1); va.push_back(-ys); }
            if(i == 0) { ci.push_back(n); va.push_back(T{1.0}); }  // B column of the source at node 0
        }
        else
        {
            ci.push_back(0); va.push_back(T{1.0});  // C row: v(0) = E
        }
        rp.push_back(static_cast<int>(ci.size()));
    }
}

template <class T>
static std::vector<T> matvec(std::vector<int> const& rp, std::vector<int> const& ci, std::vector<T> const& va, std::vector<T> const& x)
{
    std::vector<T> b(x.size());
    for(size_t i = 0; i + 1 < rp.size(); ++i)
    {
        T acc{};
        for(int e = rp[i]; e < rp[i + 1]; ++e) acc += va[e] * x[ci[e]];
        b[i] = acc;
    }
    return b;
}

int main()
{
    phy_engine::solver::hip_sparse_lu solver;
    if(!solver.is_available())
    {
        std::fprintf(stderr, "no device: %s\n", pe_hip_last_error(nullptr));
        return 2;
    }
    int const n = 300, N = n + 1;
    phy_engine::solver::hip_sparse_lu::timings tm{};
    // ---- all-real
    {
        std::vector<int> rp, ci;
        std::vector<double> va;
        ladder<double>(n, 1e-3, 2e-5, rp, ci, va);
        std::vector<double> xt(N), x(N);
        for(int i = 0; i < N; ++i) xt[i] = std::sin(0.37 * i) + 2.0;
        auto b = matvec(rp, ci, va, xt);
        if(!solver.solve_csr_real(N, static_cast<int>(ci.size()), rp.data(), ci.data(), va.data(), b.data(), x.data(), tm, true)) return 10;
        for(int i = 0; i < N; ++i)
            if(!(std::fabs(x[i] - xt[i]) <= 1e-9 * (1.0 + std::fabs(xt[i])))) { std::fprintf(stderr, "real: x[%d] = %.17g, want %.17g\n", i, x[i], xt[i]); return 11; }
        for(auto& v: va) v *= 4.0;  // cached pattern, new values
        if(!solver.solve_csr_real(N, static_cast<int>(ci.size()), rp.data(), ci.data(), va.data(), b.data(), x.data(), tm, false)) return 12;
        for(int i = 0; i < N; ++i)
            if(!(std::fabs(4.0 * x[i] - xt[i]) <= 1e-9 * (1.0 + std::fabs(xt[i])))) return 13;
    }
    // ---- complex: conductances + j omega C, like the AC stamp of an R-C ladder (circuit.h:389-431)
    {
        std::vector<int> rp, ci;
        std::vector<cplx> va;
        ladder<cplx>(n, cplx{1e-3, 0.0}, cplx{1e-6, 3e-4}, rp, ci, va);
        std::vector<cplx> xt(N), x(N);
        for(int i = 0; i < N; ++i) xt[i] = cplx{std::cos(0.21 * i) + 1.5, std::sin(0.13 * i) - 0.25};
        auto b = matvec(rp, ci, va, xt);
        if(!solver.solve_csr_timed(N, static_cast<int>(ci.size()), rp.data(), ci.data(), va.data(), b.data(), x.data(), tm, true)) return 20;
        for(int i = 0; i < N; ++i)
            if(!(std::abs(x[i] - xt[i]) <= 1e-9 * (1.0 + std::abs(xt[i])))) { std::fprintf(stderr, "complex: x[%d] = (%.17g, %.17g), want (%.17g, %.17g)\n", i, x[i].real(), x[i].imag(), xt[i].real(), xt[i].imag()); return 21; }
        if(!(tm.solve_ms >= 0.0 && tm.solve_total_host_ms > 0.0)) return 22;
        // another frequency on the cached pattern (the susceptances scale, the conductances stay): solve_csr, the untimed wrapper
        for(auto& v: va) v = cplx{v.real(), 50.0 * v.imag()};
        b = matvec(rp, ci, va, xt);
        if(!solver.solve_csr(N, static_cast<int>(ci.size()), rp.data(), ci.data(), va.data(), b.data(), x.data(), false)) return 23;
        for(int i = 0; i < N; ++i)
            if(!(std::abs(x[i] - xt[i]) <= 1e-9 * (1.0 + std::abs(xt[i])))) return 24;
        // purely imaginary diagonal: every Ar diagonal entry is zero, the matching has to pair row i with column n + i
        for(auto& v: va) v = cplx{0.0, v.real() + v.imag()};
        b = matvec(rp, ci, va, xt);
        if(!solver.solve_csr_timed(N, static_cast<int>(ci.size()), rp.data(), ci.data(), va.data(), b.data(), x.data(), tm, true)) return 25;
        for(int i = 0; i < N; ++i)
            if(!(std::abs(x[i] - xt[i]) <= 1e-9 * (1.0 + std::abs(xt[i])))) return 26;
    }
    // ---- a singular complex system comes back false (reference: solve_once returns false, circuit.h:1517)
    {
        std::vector<int> rp{0, 2, 4}, ci{0, 1, 0, 1};
        std::vector<cplx> va{{1.0, 1.0}, {2.0, 2.0}, {2.0, 2.0}, {4.0, 4.0}}, b{{1.0, 0.0}, {0.0, 1.0}}, x(2);
        if(solver.solve_csr_timed(2, 4, rp.data(), ci.data(), va.data(), b.data(), x.data(), tm, true)) return 30;
    }
    std::puts("hip_sparse_lu seam: ok");
    return 0;
}
