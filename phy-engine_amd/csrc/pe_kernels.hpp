// pe_kernels.hpp -- host-callable launchers of pe_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

#include "pe_device.hpp"

#ifndef PE_THREADS
    #define PE_THREADS 512  // upper bound (launch bounds); a launch uses V.n_waves * 64 threads
#endif

namespace pe
{
    hipError_t launch_tr_steps(hipStream_t st, DevView const& V, double dt, int nsteps, bool reuse_factor);
    hipError_t launch_dc_point(hipStream_t st, DevView const& V, int mode);
    hipError_t launch_factor_solve(hipStream_t st, DevView const& V, bool do_factor);
    // multi-workgroup mode (V.n_parts > 1)
    hipError_t launch_m2_companion(hipStream_t st, DevView const& V, double dt);
    // stamp_mode: 0 everything; 1 not the first Newton iteration of this solve point -- only the x-dependent slots / rows are stamped again (V.dyn_a /
    // dyn_b); 2 first iteration of a transient step whose matrix holds the stamp of this dt: x-dependent matrix slots + the whole right-hand side
    // companion: first iteration of a transient step -- the companion update of that step (dt = companion_dt) runs inside the evaluation launch
    hipError_t launch_m2_iteration(hipStream_t st, DevView const& V, int mode, double t, double last_step, bool do_factor, hipEvent_t ev0 = nullptr,
                                   hipEvent_t ev1 = nullptr, int stamp_mode = 0, bool companion = false, double companion_dt = 0.0);
    // the same sequence + the publication of its results (launch_m2_publish) as ONE captured graph launch, built on first use per (mode,
    // do_factor, stamp_dynamic, companion, V) and replayed afterwards; small sweeps only (pe_engine_newton.cpp).  No HIP events around the
    // dominant launch in this path.  The cache belongs to an engine (created / destroyed with it, cleared when its view changes for good).
    struct M2GraphCache;
    M2GraphCache* m2_graphs_create();
    void m2_graphs_destroy(M2GraphCache* c);
    void m2_graphs_clear(M2GraphCache* c);
    hipError_t launch_m2_iteration_graph(hipStream_t st, M2GraphCache* cache, DevView const& V, int mode, double t, double last_step, bool do_factor, int stamp_mode,
                                         bool companion, double companion_dt, int* pub_flags, double* pub_eta, unsigned long long* pub_seq, unsigned long long seq);
    // device-to-device stream copy of `bytes` (multiple of 16): the kernel behind pe_hip_measure_hbm_ceiling
    hipError_t launch_stream_copy(hipStream_t st, void const* src, void* dst, size_t bytes);
    // one round of iterative refinement of the active instances' last solve + re-check (residual safety net, pe_front.hpp)
    hipError_t launch_m2_refine(hipStream_t st, DevView const& V);
    // flags + residual norms of the iteration just launched -> pinned host memory, then `seq` into *pub_seq (device-visible pointers)
    hipError_t launch_m2_publish(hipStream_t st, DevView const& V, int* pub_flags, double* pub_eta, unsigned long long* pub_seq, unsigned long long seq);
    // small-signal AC refinement on the device (pe_front.hpp ac_residual): r = b0 - A xacc of every instance into its right-hand-side
    // value slots dv[rhs0 ..), *worst (device, one double) = max componentwise backward error over all instances
    hipError_t launch_ac_residual(hipStream_t st, DevView const& V, double const* xacc, double const* b0, int rhs0, double* worst);
    // xacc = first ? x : xacc + x  (every instance; x = the engine's current solution);  first also keeps b0 = the stamped right-hand side
    hipError_t launch_ac_accumulate(hipStream_t st, DevView const& V, double* xacc, double* b0, bool first);
    // complex twin of the solver seam (batch 1, aval in CSR order, V.csr_rp / V.csr_ci set): V.rhs = b0 - A xacc, *worst (device, one double) = max
    // componentwise backward error
    hipError_t launch_csr_residual(hipStream_t st, DevView const& V, double const* xacc, double const* b0, double* worst);
    // per-row {sum v, sum v^2, min, max} of x over the instances (the payload of the sweep's one exchange step):
    // `partial` holds n_chunks x 4 x rows doubles, `out` 4 x rows (both device memory); deterministic (fixed chunk order)
    hipError_t launch_sweep_statistics(hipStream_t st, DevView const& V, int n_chunks, double* partial, double* out);
}  // namespace pe
