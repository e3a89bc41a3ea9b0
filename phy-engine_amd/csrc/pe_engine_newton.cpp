// pe_engine_newton.cpp -- the host-driven schedule: Newton and transient loops of circult::solve / analyze (circuit.h:892-985, 233-256)
// over the per-phase kernels of the split schedule, the host-stamp overlay callbacks, the residual safety net of the static-pivot LU,
// and the two analysis entry points pe_hip_analyze_tr / pe_hip_analyze_dc (resident kernel or split schedule).
#include "pe_engine_internal.hpp"

using namespace pe_eng;

namespace pe_eng PE_ENG_HIDDEN
{
    // host-stamp overlay: one callback (+ the upload of its values for ITERATE) on the current x of instance b.  In a batch the
    // callback is told first which instance the calls that follow concern (PE_HIP_OVERLAY_INSTANCE): models with state of their own
    // (a junction's last voltage, a companion history) keep one copy per instance.
    int overlay_call(pe_hip_engine* h, int event, int mode, double t, double dt, int b)
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
            // captured launch sequences (m2_point: graph mode) keep ONE grid: the quads of the full sweep, the ones behind the active
            // instances empty (the lane-group kernels return on a first member < 0)
            if(h->graph_mode)
                for(int const nq_all = static_cast<int>((B + 3) / 4); nq < nq_all; ++nq)
                    for(int c = 0; c < 4; ++c) ql[4 * nq + c] = -1;
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

    // The same without a copy command: the iteration's last launch (k_m2_publish) writes flags + residual norms into pinned host memory
    // and then a sequence number; the host polls that word.  Falls back to the stream's completion if the word does not arrive (a failed
    // launch): the caller then sees the error of the synchronisation.
    struct Published
    {
        unsigned long long* seq;
        int* flags;
        double* eta;
    };
    Published pub_view(void* base, size_t B)
    {
        auto* p = static_cast<char*>(base);
        return {reinterpret_cast<unsigned long long*>(p), reinterpret_cast<int*>(p + 64), reinterpret_cast<double*>(p + 64 + ((B * sizeof(int) + 63) / 64) * 64)};
    }
    int ensure_published(pe_hip_engine* h, size_t B)
    {
        size_t const need = 64 + ((B * sizeof(int) + 63) / 64) * 64 + 4 * B * sizeof(double);
        if(h->pub_cap >= need) return PE_HIP_OK;
        if(h->pub_host) (void)hipHostFree(h->pub_host);
        h->pub_host = h->pub_dev = nullptr;
        h->pub_cap = 0;
        // (coherent + mapped, explicitly: the device's system-scope release of the sequence word must reach the polling host even where
        //  HIP_HOST_COHERENT=0 makes default pinned allocations non-coherent)
        HIPCHK(h, hipHostMalloc(&h->pub_host, need, hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(h, hipHostGetDevicePointer(&h->pub_dev, h->pub_host, 0));
        std::memset(h->pub_host, 0, need);
        h->pub_cap = need;
        return PE_HIP_OK;
    }
    // waits for the publication with sequence number `seq` (launched by the caller: launch_m2_publish, or the last node of a captured
    // iteration); `eta` (4 doubles per instance) may be null
    int wait_published(pe_hip_engine* h, std::vector<int>& flags, std::vector<double>* eta, unsigned long long seq)
    {
        size_t const B = flags.size();
        auto const host = pub_view(h->pub_host, B);
        for(unsigned spins = 0; __atomic_load_n(host.seq, __ATOMIC_ACQUIRE) != seq; ++spins)
        {
            if((spins & 1023u) == 1023u)
            {
                hipError_t const q = hipStreamQuery(h->stream);
                if(q == hipSuccess)
                {
                    if(__atomic_load_n(host.seq, __ATOMIC_ACQUIRE) == seq) break;
                    return fail(h, PE_HIP_ERR_INTERNAL, "the iteration's results were not published (k_m2_publish did not run)");
                }
                if(q != hipErrorNotReady) HIPCHK(h, q);
                if(spins > (1u << 16)) std::this_thread::yield();  // (a long iteration of a large sweep: leave the core to others)
            }
        }
        std::copy(host.flags, host.flags + B, flags.begin());
        if(eta && h->V.residual_tol > 0.0) eta->assign(host.eta, host.eta + 4 * B);
        return PE_HIP_OK;
    }
    // launches the publication of the iteration just enqueued and waits for it
    int publish_and_wait(pe_hip_engine* h, std::vector<int>& flags, std::vector<double>* eta)
    {
        size_t const B = flags.size();
        if(int const rc = ensure_published(h, B); rc != PE_HIP_OK) return rc;
        auto const dev = pub_view(h->pub_dev, B);
        unsigned long long const seq = ++h->pub_seq;
        HIPCHK(h, pe::launch_m2_publish(h->stream, h->V, dev.flags, dev.eta, dev.seq, seq));
        return wait_published(h, flags, eta, seq);
    }

    // Residual safety net on the host-driven schedule.  The iteration just launched left the four norms of every active instance's
    // solve in eta_acc.  Instances above the tolerance get up to two rounds of iterative refinement (launch_m2_refine: active =
    // exactly those); their flags are then the Newton / finiteness bits of the corrected x.  What refinement cannot repair leaves
    // the iteration as PE_HIP_ERR_INACCURATE (the caller re-matches on that instance's values and retries the step).
    // `published`: the norms of this iteration as k_m2_publish handed them over (empty: read them from the device)
    int m2_check_residuals(pe_hip_engine* h, M2State& S, std::vector<int>& result, int& n_active, std::vector<double> const& published)
    {
        int const B = h->hc.batch;
        std::vector<double> eta(static_cast<size_t>(B) * 4);
        bool first = published.size() == eta.size();
        auto pull_eta = [&]() -> int
        {
            if(first)  // (the iteration's own norms came with its flags; later calls follow a refinement launch)
            {
                eta = published;
                first = false;
                return PE_HIP_OK;
            }
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
    // companion_dt != null: a transient step -- its companion update rides in the first iteration's evaluation launch (k_m2_eval)
    // a_static_ok: the non-x-dependent part of every active instance's matrix already holds the stamp of this (mode, dt) -- the first iteration
    // stamps the x-dependent slots + the whole right-hand side only (stamp mode 2)
    int m2_point(pe_hip_engine* h, M2State& S, int mode, double t, double last_step, bool do_factor, std::vector<int>& result, int& launches,
                 double const* companion_dt = nullptr, bool a_static_ok = false)
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
        // Small sweeps replay a captured launch sequence per iteration (pe_kernels.hip launch_m2_iteration_graph): the 15-20 launches of an
        // iteration cost more host time than some of them run.  Large sweeps keep the plain launches -- their iteration is milliseconds,
        // and the HIP events around the dominant pair (bench.py's roofline) live there.  Knob GRAPH = 0 / 1 forces either.
        static bool const use_publish = env_int0("PHY_ENGINE_HIP_PUBLISH", 1) != 0;
        // (measured, profiles/r04_ab_runs.log: -3.3 % per iteration for the single circuit, -0.8 % at 16 instances, nothing from 128 on --
        //  there the kernels are long enough for the host to stay ahead of the stream)
        bool const graph_mode = use_publish && knob(h, "GRAPH", B <= 32 ? 1 : 0) != 0;
        if(graph_mode != h->graph_mode)
        {
            h->graph_mode = graph_mode;
            h->active_dev.clear();  // (the quad list behind the mask is laid out differently)
        }
        if(graph_mode && !h->graphs) h->graphs = pe::m2_graphs_create();
        for(int it = 0; it < max_it && n_active > 0; ++it)
        {
            if(has_overlay(h))
                if(int const rc = overlay_call_all(h, PE_HIP_OVERLAY_ITERATE, mode, t, last_step, &S.active); rc != PE_HIP_OK) return rc;
            if(int const urc = upload_active(h, S.active); urc != PE_HIP_OK) return urc;
            // (test knob PHY_ENGINE_HIP_FULL_STAMP=1: every iteration stamps everything -- the x-dependent-only path must match it bit for bit)
            static bool const full_stamp = env_int0("PHY_ENGINE_HIP_FULL_STAMP", 0) != 0;
            int const dyn = full_stamp ? 0 : (it > 0 ? 1 : (a_static_ok ? 2 : 0));
            bool const comp = it == 0 && companion_dt != nullptr;
            std::vector<double> eta_now;
            if(graph_mode)
            {
                if(int const prc = ensure_published(h, static_cast<size_t>(B)); prc != PE_HIP_OK) return prc;
                auto const dev = pub_view(h->pub_dev, static_cast<size_t>(B));
                unsigned long long const seq = ++h->pub_seq;
                HIPCHK(h, pe::launch_m2_iteration_graph(h->stream, h->graphs, h->V, mode, t, last_step, do_factor, dyn, comp, companion_dt ? *companion_dt : 0.0, dev.flags,
                                                        dev.eta, dev.seq, seq));
                ++launches;
                if(int const prc = wait_published(h, S.flags, &eta_now, seq); prc != PE_HIP_OK) return prc;
            }
            else
            {
                HIPCHK(h, pe::launch_m2_iteration(h->stream, h->V, mode, t, last_step, do_factor, h->evk0, h->evk1, /*stamp_mode=*/dyn, /*companion=*/comp,
                                                  companion_dt ? *companion_dt : 0.0));
                ++launches;
                // flags + residual norms of this iteration: published into pinned host memory by the iteration's last launch and polled
                // (no copy command, no stream synchronisation); PHY_ENGINE_HIP_PUBLISH=0: the copy + synchronise of rounds 1-2
                if(use_publish)
                {
                    if(int const prc = publish_and_wait(h, S.flags, &eta_now); prc != PE_HIP_OK) return prc;
                }
                else if(int const drc = download_flags(h, S.flags); drc != PE_HIP_OK)
                    return drc;  // (synchronises the stream)
                float kms = 0.f;
                if(hipEventElapsedTime(&kms, h->evk0, h->evk1) == hipSuccess)
                {
                    h->dominant_ms += kms;
                    ++h->dominant_launches;
                }
            }
            if(h->V.residual_tol > 0.0)
                if(int const rrc = m2_check_residuals(h, S, result, n_active, eta_now); rrc != PE_HIP_OK) return rrc;
            for(int b = 0; b < B; ++b)
            {
                if(!S.active[b]) continue;
                int const f = S.flags[b];
                if(f & 8) return fail(h, PE_HIP_ERR_INTERNAL, "a front's LDS layout exceeds the LDS of the launch that ran it (launch plan / layout mismatch)");
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
            // (the companion update of this step runs inside the first iteration's evaluation launch: m2_point / k_m2_eval -- one launch
            //  less per step; knob COMPANION_LAUNCH=1 keeps it a launch of its own)
            bool const with_companion = !(skip_first && s == 0);
            static bool const own_launch = env_int0("PHY_ENGINE_HIP_COMPANION_LAUNCH", 0) != 0;
            if(with_companion && own_launch) HIPCHK(h, pe::launch_m2_companion(h->stream, h->V, dt));
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
            // The matrix of a transient step differs from the last step's only in its x-dependent slots while dt and the parameters stay (the
            // companion conductances 2C/dt, 2L/dt are constants of the step size): instances whose matrix holds a full stamp of this dt skip the
            // other 97 % of the matrix gather on the first iteration too (k_m2_stamp 0.56 -> 0.2 ms per time point at 1 024 instances).  The
            // set is cleared wherever a static value can change (a_static_invalidate: parameters, options, load, checkpoint, re-analysis,
            // any OP / DC / TROP solve -- they stamp other companions); knob STATIC_A=0: always the full stamp.
            static bool const static_a = env_int0("PHY_ENGINE_HIP_STATIC_A", 1) != 0;
            bool a_static_ok = static_a && h->a_static_dt == dt && h->a_static.size() == static_cast<size_t>(B);
            for(int b = 0; b < B && a_static_ok; ++b)
                if(S.active[b] && !h->a_static[static_cast<size_t>(b)]) a_static_ok = false;
            rc = m2_point(h, S, PE_HIP_MODE_TR, t, dt, !reuse, res, launches, (with_companion && !own_launch) ? &dt : nullptr, a_static_ok);
            if(rc == PE_HIP_OK && !a_static_ok)
            {
                // the first iteration of this step stamped everything for the instances that were active at its start (S.active is spent by now:
                // the mask of the step is status OK && (!only || only[b]) -- recomputed)
                if(h->a_static_dt != dt || h->a_static.size() != static_cast<size_t>(B)) h->a_static.assign(static_cast<size_t>(B), 0);
                h->a_static_dt = dt;
                for(int b = 0; b < B; ++b)
                    if(res[b] != 0) h->a_static[static_cast<size_t>(b)] = 1;  // (res != 0: the instance took part in this point)
            }
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
        h->a_static.clear();  // (an OP / DC / TROP stamp overwrites the transient companions in the matrix)
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
}  // namespace pe_eng

extern "C" {

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
    if(done < nsteps) h->a_static.clear();  // (the resident kernel stamps the matrix itself: what the split schedule knew about it is void)
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
    h->a_static.clear();  // (an OP / DC / TROP solve stamps other companion values into the matrix, on either schedule)
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

}  // extern "C"
