// pe_engine.cpp -- C ABI (include/pe_hip.h) and host orchestration of the resident transient path.
//
// The host side does what circult::analyze()/prepare() do around the hot loop (circuit.h:179-296, 468-890):
// index, build the pattern once, decide how many steps to run; everything per time step runs in the kernels.
// There is deliberately NO CPU numeric fallback here: without a HIP device every compute entry point fails
// with PE_HIP_ERR_NO_DEVICE and a message.
// (the other translation units of the engine: pe_engine_internal.hpp)
#include "pe_engine_internal.hpp"

using namespace pe_eng;

namespace pe_eng PE_ENG_HIDDEN
{
    thread_local std::string g_create_error;

    int fail(pe_hip_engine* h, int code, std::string msg)
    {
        h->err = std::move(msg);
        return code;
    }

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
// device side of a load: uploads h->hc (topology, contribution lists, parameters) and allocates the per-instance state
// The slots of the device value vector that are written at LOAD time only (1.0, g_min, conductances, DC sources, digital drives, the
// static generator values) from the host circuit's CURRENT parameters -- also re-applied over a restored checkpoint, whose blob holds
// the saver's copy of them (pe_hip_checkpoint_load: a parameter edited since, or a circuit built with other values, keeps its own).
void fill_static_dv(pe_hip_engine const* h, std::vector<double>& dv)
{
    auto const& hc = h->hc;
    size_t const B = static_cast<size_t>(hc.batch);
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
}

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
        // ... and a pinned landing buffer for its 4 x rows result: the first asynchronous copy into PAGEABLE host memory makes the runtime
        // set up its staging path -- 7.4 ms on the first pe_hip_sweep_statistics call against 0.07 ms on the following ones (round 4,
        // scripts/r4_reduce_probe.py); that call is the sweep's one exchange step and sits on every rank's critical path before the all-reduce
        size_t const out_bytes = static_cast<size_t>(4) * hc.rows * sizeof(double);
        if(h->stats_pinned_bytes < out_bytes)
        {
            if(h->stats_pinned) (void)hipHostFree(h->stats_pinned);
            h->stats_pinned = nullptr;
            h->stats_pinned_bytes = 0;
            HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->stats_pinned), std::max<size_t>(out_bytes, 8), hipHostMallocDefault));
            h->stats_pinned_bytes = out_bytes;
        }
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
        fill_static_dv(h, dv);
        double* ddv{};
        HIPCHK(h, P.alloc(ddv, dv.size(), false));
        HIPCHK(h, hipMemcpy(ddv, dv.data(), dv.size() * sizeof(double), hipMemcpyHostToDevice));
        V.dv = ddv;
    }
    apply_options(h, V);
    h->V = V;
    h->loaded = true;
    // One dry run of the sweep's exchange step (statistics kernels + the asynchronous copy of their result into the pinned buffer) on the
    // zeroed solution: the first use of that path costs the runtime 7-8 ms (first asynchronous device-to-host copy of the stream: copy-queue
    // set-up; measured with scripts/r4_reduce_probe.py: 7.4 ms, then 0.07 ms) -- paid here, with the load, instead of on every rank's
    // critical path in front of the all-reduce.
    if(hc.rows > 0 && h->stats_pinned)
    {
        int const n_chunks = stats_chunks(hc.batch);
        double* dev_out = h->stats_scratch + static_cast<size_t>(n_chunks) * 4 * hc.rows;
        HIPCHK(h, pe::launch_sweep_statistics(h->stream, h->V, n_chunks, h->stats_scratch, dev_out));
        HIPCHK(h, hipMemcpyAsync(h->stats_pinned, dev_out, static_cast<size_t>(4) * hc.rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PE_HIP_OK;
}
}  // namespace pe_eng

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
    if(h->graphs) pe::m2_graphs_destroy(h->graphs);  // (captured launch sequences: before the memory they point into and their stream go)
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
    if(h->pub_host) (void)hipHostFree(h->pub_host);
    if(h->stats_pinned) (void)hipHostFree(h->stats_pinned);
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
        h->a_static.clear();
    }
    if(h->loaded && gmin_changed)
    {
        HIPCHK(h, hipSetDevice(h->device));
        std::vector<double> col(h->hc.batch, o->g_min);
        HIPCHK(h, hipMemcpy2D(h->V.dv + pe::DV_GMIN, h->hc.dv_len * sizeof(double), col.data(), sizeof(double), sizeof(double), h->hc.batch,
                              hipMemcpyHostToDevice));
        h->fact_valid = false;
        h->a_static.clear();
    }
    return PE_HIP_OK;
}

int pe_hip_set_knob(pe_hip_engine* h, const char* name, int value)
{
    if(!h || !name || !*name) return PE_HIP_ERR_ARG;
    std::string key{name};
    if(key.rfind("PHY_ENGINE_HIP_", 0) == 0) key.erase(0, 15);
    if(key.empty()) return fail(h, PE_HIP_ERR_ARG, "set_knob: empty name");
    h->knobs[key] = value;
    h->sym_class = -1;  // the launch geometry and the symbolic analysis are chosen again at the next analysis
    return PE_HIP_OK;
}

int pe_hip_get_knob(pe_hip_engine* h, const char* name, int* value, int* is_set)
{
    if(!h || !name || !value) return PE_HIP_ERR_ARG;
    std::string key{name};
    if(key.rfind("PHY_ENGINE_HIP_", 0) == 0) key.erase(0, 15);
    auto const it = h->knobs.find(key);
    char const* env = std::getenv(("PHY_ENGINE_HIP_" + key).c_str());
    bool const set = it != h->knobs.end() || (env && *env);
    if(is_set) *is_set = set ? 1 : 0;
    *value = it != h->knobs.end() ? it->second : (env && *env ? std::atoi(env) : 0);
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
                h->hc.drv_volt[k] = volt[k];  // (the host circuit mirrors the resident values: fill_static_dv)
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
    h->a_static.clear();
    pe::m2_graphs_clear(h->graphs);  // (captured launch sequences point into the circuit being replaced)
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
        out->mid_top_limit = h->V.mid_top_limit;
        out->ew_grid = h->V.ew_grid;
        out->quad_lds_pad = h->V.quad_lds_pad;
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
    h->a_static.clear();
    return PE_HIP_OK;
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
    size_t const out_bytes = static_cast<size_t>(4) * rows * sizeof(double);
    bool const pinned = h->stats_pinned && h->stats_pinned_bytes >= out_bytes;
    HIPCHK(h, hipMemcpyAsync(pinned ? static_cast<void*>(h->stats_pinned) : static_cast<void*>(out), dev_out, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if(pinned) std::memcpy(out, h->stats_pinned, out_bytes);
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
        h->a_static.clear();
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
    h->a_static.clear();
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

// (pe_build_id.hpp is generated by csrc/Makefile; the host emulation of tests/emu is not a build of the library and says so)
#if __has_include("pe_build_id.hpp")
    #include "pe_build_id.hpp"
#elif defined(PE_REQUIRE_BUILD_ID)
    #error "pe_build_id.hpp missing: build through phy-engine_amd/csrc/Makefile"
#else
    #define PE_BUILD_ID "host-emulation"
#endif
const char* pe_hip_build_id(void) { return PE_BUILD_ID; }

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
        if(char const* bad = csr_pattern_error(n, nnz, row_ptr, col_ind)) return fail(h, PE_HIP_ERR_ARG, std::string("solve_csr_real: ") + bad);
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
