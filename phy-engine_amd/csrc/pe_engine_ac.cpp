// pe_engine_ac.cpp -- small-signal AC on the device: pe_hip_analyze_ac / pe_hip_get_solution_ac (the real-equivalent system of pe_ac.hpp
// solved by a second engine, iterative refinement on the device).
#include "pe_engine_internal.hpp"

using namespace pe_eng;

extern "C" {

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
        A.eng->knobs = h->knobs;  // (the real-equivalent system is analysed under the same tuning knobs)
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
    auto const& ah = A.circ.hc;
    std::vector<double> dv(static_cast<size_t>(B) * ah.dv_len);
    if(has_overlay(h))
    {
        h->ov_x.resize(static_cast<size_t>(hc.rows));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    for(int b = 0; b < B; ++b)
    {
        if(has_overlay(h))
        {
            // host-stamped models: their iterate_ac hooks stamp complex values at this omega around the operating point held in x -- per
            // instance, as the transient path does (PE_HIP_OVERLAY_INSTANCE tells the callback whose state the call concerns); the
            // reference runs every model's iterate_ac in its AC loop, circuit.h:389-431
            if(B > 1 && h->overlay_fn(h->overlay_user, PE_HIP_OVERLAY_INSTANCE, b, omega, 0.0, nullptr, nullptr, nullptr) != 0)
                return fail(h, PE_HIP_ERR_INTERNAL, "analyze_ac: host-stamp overlay: the callback refused PE_HIP_OVERLAY_INSTANCE (it does not support batches)");
            op.ov_a.assign(2 * static_cast<size_t>(hc.n_ov_a), 0.0);
            op.ov_b.assign(2 * static_cast<size_t>(hc.n_ov_b), 0.0);
            HIPCHK(h, hipMemcpy(h->ov_x.data(), h->V.x + static_cast<size_t>(b) * hc.rows, static_cast<size_t>(hc.rows) * sizeof(double), hipMemcpyDeviceToHost));
            if(h->overlay_fn(h->overlay_user, PE_HIP_OVERLAY_AC, PE_HIP_MODE_OP, omega, 0.0, h->ov_x.data(), op.ov_a.data(), op.ov_b.data()) != 0)
                return fail(h, PE_HIP_ERR_INTERNAL, "analyze_ac: host-stamp overlay: a model's iterate_ac hook failed");
        }
        pe::fill_ac_values(hc, A.circ, op, b, omega, h->opt.g_min, r_open_of(h), &dv[static_cast<size_t>(b) * ah.dv_len]);
    }
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

}  // extern "C"
