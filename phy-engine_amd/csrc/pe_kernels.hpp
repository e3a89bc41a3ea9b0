// pe_kernels.hpp -- host-callable launchers of pe_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

#include "pe_device.hpp"

#ifndef PE_THREADS
    #define PE_THREADS 256
#endif

namespace pe
{
    size_t lds_bytes_for(DevView const& V, int max_m);
    hipError_t launch_tr_steps(hipStream_t st, DevView const& V, double dt, int nsteps, bool reuse_factor, size_t lds_bytes);
    hipError_t launch_dc_point(hipStream_t st, DevView const& V, int mode, size_t lds_bytes);
    hipError_t launch_factor_solve(hipStream_t st, DevView const& V, bool do_factor, size_t lds_bytes);
}  // namespace pe
