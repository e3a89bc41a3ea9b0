// tests/emu/pe_kernels_emu.cpp -- TEST INFRASTRUCTURE ONLY: runs the team-generic code of pe_front.hpp with a
// one-thread team per instance on the host (see hip_shim/hip/hip_runtime_api.h).  Checks indexing and the
// orchestration; says nothing about races or performance.
#include <vector>

#include "pe_front.hpp"
#include "pe_kernels.hpp"

namespace pe
{
    struct SerialTeam
    {
        int tid() const { return 0; }
        int size() const { return 1; }
        void sync() const {}
        int sync_or(int v) const { return v; }
    };

    size_t lds_bytes_for(DevView const& V, int max_m) { return (static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap + static_cast<size_t>(max_m) + 2) * sizeof(double); }

    hipError_t launch_tr_steps(hipStream_t, DevView const& V, double dt, int nsteps, bool reuse, size_t lds)
    {
        std::vector<double> mem(lds / sizeof(double) + 1);
        for(int b = 0; b < V.batch; ++b)
            tr_steps(SerialTeam{}, V, b, dt, nsteps, reuse, mem.data(), mem.data() + static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap);
        return hipSuccess;
    }
    hipError_t launch_dc_point(hipStream_t, DevView const& V, int mode, size_t lds)
    {
        std::vector<double> mem(lds / sizeof(double) + 1);
        for(int b = 0; b < V.batch; ++b) dc_point(SerialTeam{}, V, b, mode, mem.data(), mem.data() + static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap);
        return hipSuccess;
    }
    hipError_t launch_factor_solve(hipStream_t, DevView const& V, bool do_factor, size_t lds)
    {
        std::vector<double> mem(lds / sizeof(double) + 1);
        SerialTeam tm;
        for(int b = 0; b < V.batch; ++b)
        {
            int st = ST_OK;
            if(do_factor && !factor_all(tm, V, b, mem.data())) st = ST_SINGULAR;
            if(st == ST_OK)
            {
                solve_all(tm, V, b, mem.data() + static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap);
                double const* x = V.x + static_cast<long long>(b) * V.rows;
                for(int r = 0; r < V.rows; ++r)
                    if(!(std::fabs(x[r]) <= 1.7976931348623157e308)) st = ST_SINGULAR;
            }
            V.status[b] = st;
        }
        return hipSuccess;
    }
}  // namespace pe
