// pe_sweep.cpp -- Monte-Carlo / parameter sweep over the GPUs of one node, behind the C ABI (include/pe_hip.h "sweep").
//
// independent circuit instances are the natural shard -- identical topology, so the symbolic analysis is replicated
// per device; the per-instance data is partitioned in CONTIGUOUS blocks of ceil(batch / G) instances per device, the chunk rule
// of the reference's only multi-device code (src/pe_synth_cuda_u64_cones.cu:1894-1904), whose calling convention this mirrors:
// an extern "C" entry point that takes a device mask (:1861-1872).  One engine and one host thread per device while a call runs;
// no data-path exchange between devices -- the only combination step is the reduction of the per-row statistics at the end,
// done here on the host in device order (bitwise reproducible; bench.py's one-process-per-GPU launch does the same reduction
// with two RCCL all-reduces).
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pe_hip.h"
#include "pe_circuit.hpp"  // gen_ncol: parameter columns of the generic device kinds

struct pe_hip_sweep
{
    std::vector<int> device;
    std::vector<pe_hip_engine*> eng;
    std::vector<int> lo, hi;  // instances [lo, hi) of the whole sweep on engine g
    int batch{}, rows{};
    bool loaded{};
    std::string err;
};

namespace
{
    thread_local std::string g_sweep_create_error;

    int param_columns(int kind)
    {
        if(kind == PE_HIP_VAC) return 3;
        if(kind == PE_HIP_DIODE) return PE_HIP_DIODE_NPARAM;
        return pe::gen_ncol(kind);
    }

    // fn(g) on one host thread per engine that holds instances; the first failing engine's code and message are kept
    template <class F>
    int for_each_engine(pe_hip_sweep* s, F&& fn)
    {
        size_t const G = s->eng.size();
        std::vector<int> rc(G, PE_HIP_OK);
        std::vector<std::thread> th;
        for(size_t g = 0; g < G; ++g)
            if(s->hi[g] > s->lo[g]) th.emplace_back([&, g] { rc[g] = fn(static_cast<int>(g)); });
        for(auto& t: th) t.join();
        for(size_t g = 0; g < G; ++g)
            if(rc[g] != PE_HIP_OK)
            {
                s->err = "device " + std::to_string(s->device[g]) + ": " + pe_hip_last_error(s->eng[g]);
                return rc[g];
            }
        return PE_HIP_OK;
    }
}  // namespace

extern "C" {

int pe_hip_sweep_create(unsigned device_mask, pe_hip_sweep** out)
{
    if(!out) return PE_HIP_ERR_ARG;
    *out = nullptr;
    int const n = pe_hip_device_count();
    if(n <= 0)
    {
        g_sweep_create_error = "no HIP device visible: the MI355X engine has no CPU fallback";
        return PE_HIP_ERR_NO_DEVICE;
    }
    auto s = std::make_unique<pe_hip_sweep>();
    for(int d = 0; d < 32; ++d)
        if(device_mask & (1u << d))
        {
            if(d >= n)
            {
                g_sweep_create_error = "device mask names device " + std::to_string(d) + ", " + std::to_string(n) + " visible";
                for(auto* e: s->eng) pe_hip_destroy(e);
                return PE_HIP_ERR_ARG;
            }
            pe_hip_engine* e = nullptr;
            if(int const rc = pe_hip_create(d, &e); rc != PE_HIP_OK)
            {
                g_sweep_create_error = pe_hip_last_error(nullptr);
                for(auto* q: s->eng) pe_hip_destroy(q);
                return rc;
            }
            s->device.push_back(d);
            s->eng.push_back(e);
        }
    if(s->eng.empty())
    {
        g_sweep_create_error = "empty device mask";
        return PE_HIP_ERR_ARG;
    }
    s->lo.assign(s->eng.size(), 0);
    s->hi.assign(s->eng.size(), 0);
    *out = s.release();
    return PE_HIP_OK;
}

void pe_hip_sweep_destroy(pe_hip_sweep* s)
{
    if(!s) return;
    for(auto* e: s->eng) pe_hip_destroy(e);
    delete s;
}

const char* pe_hip_sweep_last_error(pe_hip_sweep* s) { return s ? s->err.c_str() : g_sweep_create_error.c_str(); }

int pe_hip_sweep_devices(pe_hip_sweep* s) { return s ? static_cast<int>(s->eng.size()) : 0; }

int pe_hip_sweep_shard(pe_hip_sweep* s, int index, int* device, int* first_instance, int* count)
{
    if(!s || index < 0 || index >= static_cast<int>(s->eng.size())) return PE_HIP_ERR_ARG;
    if(device) *device = s->device[index];
    if(first_instance) *first_instance = s->lo[index];
    if(count) *count = s->hi[index] - s->lo[index];
    return PE_HIP_OK;
}

int pe_hip_sweep_set_options(pe_hip_sweep* s, const pe_hip_options* opt)
{
    if(!s || !opt) return PE_HIP_ERR_ARG;
    for(size_t g = 0; g < s->eng.size(); ++g)
        if(int const rc = pe_hip_set_options(s->eng[g], opt); rc != PE_HIP_OK)
        {
            s->err = pe_hip_last_error(s->eng[g]);
            return rc;
        }
    return PE_HIP_OK;
}

int pe_hip_sweep_load_circuit(pe_hip_sweep* s, int n_nodes, int n_branches, int batch, int n_tables, const pe_hip_device_table* tables)
{
    if(!s || batch <= 0 || n_tables < 0 || (n_tables > 0 && !tables)) return PE_HIP_ERR_ARG;
    int const G = static_cast<int>(s->eng.size());
    int const chunk = (batch + G - 1) / G;  // contiguous blocks of ceil(batch / G): pe_synth_cuda_u64_cones.cu:1894-1904
    for(int g = 0; g < G; ++g)
    {
        s->lo[g] = std::min(batch, g * chunk);
        s->hi[g] = std::min(batch, s->lo[g] + chunk);
    }
    s->batch = batch;
    s->rows = n_nodes + n_branches;
    s->loaded = false;
    int const rc = for_each_engine(s,
                                   [&](int g)
                                   {
                                       // the shard's view of the tables: batched parameter blocks start at this device's first instance
                                       std::vector<pe_hip_device_table> t(tables, tables + n_tables);
                                       for(auto& q: t)
                                           if(q.params_batched && q.params) q.params += static_cast<size_t>(s->lo[g]) * q.count * param_columns(q.kind);
                                       return pe_hip_load_circuit(s->eng[g], n_nodes, n_branches, s->hi[g] - s->lo[g], n_tables, t.data());
                                   });
    s->loaded = rc == PE_HIP_OK;
    return rc;
}

int pe_hip_sweep_reset(pe_hip_sweep* s)
{
    if(!s || !s->loaded) return PE_HIP_ERR_ARG;
    return for_each_engine(s, [&](int g) { return pe_hip_reset(s->eng[g]); });
}

// `nsteps` transient steps of every instance; stats (may be NULL): sums over the devices, times = the slowest device
int pe_hip_sweep_run(pe_hip_sweep* s, double dt, int nsteps, pe_hip_run_stats* stats)
{
    if(!s || !s->loaded) return PE_HIP_ERR_ARG;
    std::vector<pe_hip_run_stats> st(s->eng.size());
    int const rc = for_each_engine(s, [&](int g) { return pe_hip_analyze_tr(s->eng[g], dt, nsteps, &st[g]); });
    if(stats)
    {
        *stats = pe_hip_run_stats{};
        for(size_t g = 0; g < s->eng.size(); ++g)
        {
            if(s->hi[g] <= s->lo[g]) continue;
            stats->steps += st[g].steps;
            stats->newton_iters += st[g].newton_iters;
            stats->gpu_ms = std::max(stats->gpu_ms, st[g].gpu_ms);
            stats->n_launches = std::max(stats->n_launches, st[g].n_launches);
            stats->n_failed += st[g].n_failed;
            stats->dominant_ms = std::max(stats->dominant_ms, st[g].dominant_ms);
            stats->dominant_launches = std::max(stats->dominant_launches, st[g].dominant_launches);
        }
    }
    return rc;
}

int pe_hip_sweep_operating_point(pe_hip_sweep* s, int mode, pe_hip_run_stats* stats)
{
    if(!s || !s->loaded) return PE_HIP_ERR_ARG;
    std::vector<pe_hip_run_stats> st(s->eng.size());
    int const rc = for_each_engine(s, [&](int g) { return pe_hip_analyze_dc(s->eng[g], mode, &st[g]); });
    if(stats)
    {
        *stats = pe_hip_run_stats{};
        for(size_t g = 0; g < s->eng.size(); ++g)
        {
            if(s->hi[g] <= s->lo[g]) continue;
            stats->steps += st[g].steps;
            stats->newton_iters += st[g].newton_iters;
            stats->gpu_ms = std::max(stats->gpu_ms, st[g].gpu_ms);
            stats->n_failed += st[g].n_failed;
        }
    }
    return rc;
}

// out[4][rows]: sum, sum of squares, min, max over ALL instances -- each device reduces its block on the device
// (pe_hip_sweep_statistics), the blocks are combined here in device order
int pe_hip_sweep_reduce(pe_hip_sweep* s, double* out)
{
    if(!s || !s->loaded || !out) return PE_HIP_ERR_ARG;
    size_t const R = static_cast<size_t>(s->rows);
    std::vector<std::vector<double>> part(s->eng.size(), std::vector<double>(4 * R));
    if(int const rc = for_each_engine(s, [&](int g) { return pe_hip_sweep_statistics(s->eng[g], part[g].data()); }); rc != PE_HIP_OK) return rc;
    bool first = true;
    for(size_t g = 0; g < s->eng.size(); ++g)
    {
        if(s->hi[g] <= s->lo[g]) continue;
        double const* p = part[g].data();
        if(first) std::memcpy(out, p, 4 * R * sizeof(double));
        else
            for(size_t r = 0; r < R; ++r)
            {
                out[r] += p[r];
                out[R + r] += p[R + r];
                out[2 * R + r] = std::min(out[2 * R + r], p[2 * R + r]);
                out[3 * R + r] = std::max(out[3 * R + r], p[3 * R + r]);
            }
        first = false;
    }
    return PE_HIP_OK;
}

int pe_hip_sweep_get_solution(pe_hip_sweep* s, int first_instance, int count, double* x)
{
    if(!s || !s->loaded || !x || first_instance < 0 || count < 0 || first_instance + count > s->batch) return PE_HIP_ERR_ARG;
    for(size_t g = 0; g < s->eng.size(); ++g)
    {
        int const a = std::max(first_instance, s->lo[g]), b = std::min(first_instance + count, s->hi[g]);
        if(b <= a) continue;
        if(int const rc = pe_hip_get_solution(s->eng[g], a - s->lo[g], b - a, x + static_cast<size_t>(a - first_instance) * s->rows); rc != PE_HIP_OK)
        {
            s->err = pe_hip_last_error(s->eng[g]);
            return rc;
        }
    }
    return PE_HIP_OK;
}

int pe_hip_sweep_get_instance_state(pe_hip_sweep* s, int first_instance, int count, int* status, long long* steps, long long* iters, double* t)
{
    if(!s || !s->loaded || first_instance < 0 || count < 0 || first_instance + count > s->batch) return PE_HIP_ERR_ARG;
    for(size_t g = 0; g < s->eng.size(); ++g)
    {
        int const a = std::max(first_instance, s->lo[g]), b = std::min(first_instance + count, s->hi[g]);
        if(b <= a) continue;
        int const o = a - first_instance;
        if(int const rc = pe_hip_get_instance_state(s->eng[g], a - s->lo[g], b - a, status ? status + o : nullptr, steps ? steps + o : nullptr, iters ? iters + o : nullptr,
                                                    t ? t + o : nullptr);
           rc != PE_HIP_OK)
        {
            s->err = pe_hip_last_error(s->eng[g]);
            return rc;
        }
    }
    return PE_HIP_OK;
}

}  // extern "C"
