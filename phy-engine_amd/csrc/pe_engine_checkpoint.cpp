// pe_engine_checkpoint.cpp -- pe_hip_checkpoint_size / _save / _load (include/pe_hip.h).
#include "pe_engine_internal.hpp"

using namespace pe_eng;

extern "C" {

/* ---- checkpoint / resume of the device-resident simulation state (SURVEY.md 5 "checkpoint/resume", 8f rank 4): everything a
 * transient needs to continue bit-exactly -- solution, time, trapezoidal companion histories, junction limiting state, relay
 * contacts, counters, the device value vector -- of every instance, as one flat little-endian blob.  The circuit itself
 * (topology, parameters) is NOT in the blob: load the same circuit first; the header's sizes are checked. */
extern "C++" {
namespace
{
    struct CkHeader
    {
        char magic[8];
        long long rows, batch, nC, nD, nRl, dv_len;
        unsigned long long fingerprint;  // of the circuit's STRUCTURE in the order it was built (below): sizes alone cannot tell two
                                         // circuits with the same counts but another node / device order apart
    };
    // FNV-1a over the MNA pattern and the contribution lists of every matrix slot / right-hand-side row (which device value lands where,
    // with which sign): equal for the same netlist built in the same order, different as soon as a node or a device changes place.
    // Parameter VALUES are not part of it -- they are not taken from a blob (pe_hip_checkpoint_load re-applies the circuit's own).
    unsigned long long structure_fingerprint(pe::HostCircuit const& hc)
    {
        unsigned long long f = 1469598103934665603ull;
        auto mix = [&](void const* p, size_t n)
        {
            auto const* c = static_cast<unsigned char const*>(p);
            for(size_t i = 0; i < n; ++i) f = (f ^ c[i]) * 1099511628211ull;
        };
        auto vec = [&](std::vector<int> const& v)
        {
            size_t const n = v.size();
            mix(&n, sizeof(n));
            if(n) mix(v.data(), n * sizeof(int));
        };
        long long const sizes[8] = {hc.rows, hc.n_nodes, hc.n_branches, hc.nC(), hc.nD(), hc.nRl(), hc.dv_len, hc.nonlinear};
        mix(sizes, sizeof(sizes));
        vec(hc.rp);
        vec(hc.ci);
        vec(hc.a_ptr);
        vec(hc.a_src);
        vec(hc.b_ptr);
        vec(hc.b_src);
        return f;
    }
    struct CkPart
    {
        void* ptr;
        size_t bytes;
    };
    std::vector<CkPart> ck_parts(pe_hip_engine* h)
    {
        auto const& hc = h->hc;
        auto& V = h->V;
        size_t const B = static_cast<size_t>(hc.batch);
        return {{V.x, B * hc.rows * sizeof(double)},
                {V.t_now, B * sizeof(double)},
                {V.last_step, B * sizeof(double)},
                {V.status, B * sizeof(int)},
                {V.n_steps, B * sizeof(long long)},
                {V.n_iters, B * sizeof(long long)},
                {V.c_hist, B * hc.nC() * sizeof(double)},
                {V.c_prevg, B * hc.nC() * sizeof(double)},
                {V.d_udlast, B * hc.nD() * sizeof(double)},
                {V.d_geq, B * hc.nD() * sizeof(double)},
                {V.d_hist, B * hc.nD() * sizeof(double)},
                {V.d_prevg, B * hc.nD() * sizeof(double)},
                {V.rl_engaged, B * hc.nRl() * sizeof(int)},
                {V.dv, B * hc.dv_len * sizeof(double)}};
    }
}  // namespace
}  // extern "C++"

int pe_hip_checkpoint_size(pe_hip_engine* h, size_t* bytes)
{
    if(!h || !h->loaded || !bytes) return PE_HIP_ERR_ARG;
    size_t n = sizeof(CkHeader);
    for(auto const& p: ck_parts(h)) n += p.bytes;
    *bytes = n;
    return PE_HIP_OK;
}

int pe_hip_checkpoint_save(pe_hip_engine* h, void* buffer, size_t capacity)
{
    size_t need = 0;
    if(!buffer || pe_hip_checkpoint_size(h, &need) != PE_HIP_OK || capacity < need) return h ? fail(h, PE_HIP_ERR_ARG, "checkpoint_save: buffer too small or no circuit") : PE_HIP_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    auto const& hc = h->hc;
    CkHeader hd{{'P', 'E', 'H', 'I', 'P', 'C', 'K', '2'}, hc.rows, hc.batch, hc.nC(), hc.nD(), hc.nRl(), hc.dv_len, structure_fingerprint(hc)};
    char* o = static_cast<char*>(buffer);
    std::memcpy(o, &hd, sizeof(hd));
    o += sizeof(hd);
    for(auto const& p: ck_parts(h))
    {
        if(p.bytes) HIPCHK(h, hipMemcpy(o, p.ptr, p.bytes, hipMemcpyDeviceToHost));
        o += p.bytes;
    }
    return PE_HIP_OK;
}

int pe_hip_checkpoint_load(pe_hip_engine* h, const void* buffer, size_t size)
{
    size_t need = 0;
    if(!buffer || pe_hip_checkpoint_size(h, &need) != PE_HIP_OK || size != need) return h ? fail(h, PE_HIP_ERR_ARG, "checkpoint_load: size does not match the loaded circuit") : PE_HIP_ERR_ARG;
    auto const& hc = h->hc;
    CkHeader hd{};
    std::memcpy(&hd, buffer, sizeof(hd));
    if(std::memcmp(hd.magic, "PEHIPCK2", 8) != 0 || hd.rows != hc.rows || hd.batch != hc.batch || hd.nC != hc.nC() || hd.nD != hc.nD() || hd.nRl != hc.nRl() ||
       hd.dv_len != hc.dv_len || hd.fingerprint != structure_fingerprint(hc))
        return fail(h, PE_HIP_ERR_ARG, "checkpoint_load: the checkpoint belongs to a different circuit (sizes or structure fingerprint: another topology, or the same one built in another order)");
    HIPCHK(h, hipSetDevice(h->device));
    char const* i = static_cast<char const*>(buffer) + sizeof(hd);
    auto const parts = ck_parts(h);
    for(size_t k = 0; k + 1 < parts.size(); ++k)
    {
        if(parts[k].bytes) HIPCHK(h, hipMemcpy(parts[k].ptr, i, parts[k].bytes, hipMemcpyHostToDevice));
        i += parts[k].bytes;
    }
    // the device value vector (last part): the x-dependent / companion slots come from the blob, the slots written at load time only
    // (conductances, DC sources, g_min, ...) from THIS circuit's current parameters -- a blob must not revert a parameter edited since
    // the save, nor impose the saver's values on a circuit built with others (round-3 advisor finding)
    {
        std::vector<double> dv(static_cast<size_t>(hc.batch) * hc.dv_len);
        if(!dv.empty()) std::memcpy(dv.data(), i, dv.size() * sizeof(double));
        fill_static_dv(h, dv);
        if(!dv.empty()) HIPCHK(h, hipMemcpy(h->V.dv, dv.data(), dv.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    h->fact_valid = false;
    h->a_static.clear();
    return PE_HIP_OK;
}

}  // extern "C"
