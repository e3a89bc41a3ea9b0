// pe_kernels.hip -- gfx950 kernels of the resident transient path.
//
// Mapping (MI355X-first): ONE WORKGROUP = ONE CIRCUIT INSTANCE.  A workgroup owns its instance for a whole
// launch: companion update, device evaluation, MNA gather, multifrontal LU with the dense fronts staged in LDS,
// triangular solves with the front-local vectors in LDS, Newton test -- with workgroup barriers only, no
// inter-workgroup traffic and no host round trip inside a time step.  A Monte-Carlo sweep fills the chip with
// independent instances (256 CUs x k workgroups); symbolic data is shared read-only and stays in L2/MALL.
#include <hip/hip_runtime.h>

#include "pe_front.hpp"
#include "pe_kernels.hpp"

namespace pe
{
    struct HipTeam
    {
        __device__ __forceinline__ int tid() const { return static_cast<int>(threadIdx.x); }
        __device__ __forceinline__ int size() const { return static_cast<int>(blockDim.x); }
        __device__ __forceinline__ void sync() const { __syncthreads(); }
        __device__ __forceinline__ int sync_or(int v) const { return __syncthreads_or(v); }
    };

    // dynamic LDS: [front: cap*cap doubles][yl: max_m doubles]
    extern __shared__ __attribute__((aligned(16))) double pe_lds[];

    __global__ void __launch_bounds__(PE_THREADS) k_tr_steps(DevView V, double dt, int nsteps, int reuse_factor)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(b >= V.batch) return;
        HipTeam tm;
        double* front = pe_lds;
        double* yl = pe_lds + static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap;
        tr_steps(tm, V, b, dt, nsteps, reuse_factor != 0, front, yl);
    }

    __global__ void __launch_bounds__(PE_THREADS) k_dc_point(DevView V, int mode)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(b >= V.batch) return;
        HipTeam tm;
        double* front = pe_lds;
        double* yl = pe_lds + static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap;
        dc_point(tm, V, b, mode, front, yl);
    }

    // A x = b with A values / rhs already resident (solve_csr_real seam): factor + solve, instance 0..batch-1
    __global__ void __launch_bounds__(PE_THREADS) k_factor_solve(DevView V, int do_factor)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(b >= V.batch) return;
        HipTeam tm;
        double* front = pe_lds;
        double* yl = pe_lds + static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap;
        int st = ST_OK;
        if(do_factor && !factor_all(tm, V, b, front)) st = ST_SINGULAR;
        if(st == ST_OK)
        {
            solve_all(tm, V, b, yl);
            double const* x = V.x + static_cast<long long>(b) * V.rows;
            int nonfinite = 0;
            for(int r = tm.tid(); r < V.rows; r += tm.size())
                if(!(fabs(x[r]) <= 1.7976931348623157e308)) nonfinite = 1;
            if(tm.sync_or(nonfinite)) st = ST_SINGULAR;
        }
        if(tm.tid() == 0) V.status[b] = st;
    }

    static hipError_t set_lds(void const* fn, size_t bytes)
    {
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    }

    size_t lds_bytes_for(DevView const& V, int max_m) { return (static_cast<size_t>(V.lds_front_cap) * V.lds_front_cap + static_cast<size_t>(max_m) + 2) * sizeof(double); }

    hipError_t launch_tr_steps(hipStream_t st, DevView const& V, double dt, int nsteps, bool reuse, size_t lds)
    {
        hipError_t e = set_lds(reinterpret_cast<void const*>(&k_tr_steps), lds);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_tr_steps, dim3(V.batch), dim3(PE_THREADS), lds, st, V, dt, nsteps, reuse ? 1 : 0);
        return hipGetLastError();
    }

    hipError_t launch_dc_point(hipStream_t st, DevView const& V, int mode, size_t lds)
    {
        hipError_t e = set_lds(reinterpret_cast<void const*>(&k_dc_point), lds);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_dc_point, dim3(V.batch), dim3(PE_THREADS), lds, st, V, mode);
        return hipGetLastError();
    }

    hipError_t launch_factor_solve(hipStream_t st, DevView const& V, bool do_factor, size_t lds)
    {
        hipError_t e = set_lds(reinterpret_cast<void const*>(&k_factor_solve), lds);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_factor_solve, dim3(V.batch), dim3(PE_THREADS), lds, st, V, do_factor ? 1 : 0);
        return hipGetLastError();
    }
}  // namespace pe
