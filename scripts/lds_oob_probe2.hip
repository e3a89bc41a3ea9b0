// developer probe, round 4: the round-3 library 20478fc ran the front 118 x 40 (8 076 doubles of LDS: panels + right-hand-side column) on a launch of
// k_m2_factor_top<4> with 40 592 B of dynamic LDS -- 165 dispatches in gpurun_out/r03h_trace, no k_m2_factor_top_mid -- and the bench checksum
// was that of the fixed library, while scripts/lds_oob_probe.hip shows that gfx950 drops LDS accesses beyond a workgroup's allocation.  Both
// cannot hold for the same launch unless something about THAT launch differs from the first probe.  Variants tried here (nothing of the engine
// is involved; out-of-allocation LDS accesses do not fault):
//   A  the first probe's shape                          (no static LDS, default launch bounds, grid (1024))
//   B  the production kernel's shape                    (320 B of static LDS in front, __launch_bounds__(512, 4), 256 threads, grid (1, 1024))
//   C  as B, right after a launch of ANOTHER kernel that held 80 KB per workgroup on the same CUs (the order of the production sequence:
//      k_m2_factor_parts, then the top launches; a stale allocation size?)
//   D  as B, with the function's MaxDynamicSharedMemorySize attribute raised to 80 KB first (set_lds raises attributes, never lowers them)
// hipcc --offload-arch=gfx950 -O2 -o probe2 lds_oob_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double lds[];

__device__ void body(double* out, int n_alloc, int n_use, int wg)
{
    double const tag = 1000.0 * (wg + 1);
    for(int i = threadIdx.x; i < n_use; i += blockDim.x) lds[i] = tag + i;
    __syncthreads();
    int good_in = 0, good_out = 0, zero_out = 0;
    for(int i = threadIdx.x; i < n_use; i += blockDim.x)
    {
        double const v = lds[i];
        if(i < n_alloc) good_in += v == tag + i;
        else
        {
            good_out += v == tag + i;
            zero_out += v == 0.0;
        }
    }
    atomicAdd(out + 0, (double)good_in);
    atomicAdd(out + 1, (double)good_out);
    atomicAdd(out + 2, (double)zero_out);
}
__global__ void probe_a(double* out, int n_alloc, int n_use) { body(out, n_alloc, n_use, blockIdx.x); }
__global__ void __launch_bounds__(512, 4) probe_b(double* out, int n_alloc, int n_use)
{
    __shared__ double rd[40];  // 320 B of static LDS in front of the dynamic region, as the production kernels have
    if(threadIdx.x < 40) rd[threadIdx.x] = 1.0;
    __syncthreads();
    body(out, n_alloc, n_use, blockIdx.y);
    if(rd[threadIdx.x % 40] != 1.0) atomicAdd(out + 3, 1.0);
}
__global__ void __launch_bounds__(512, 4) hog(double* out, int n)
{
    for(int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 7.0;
    __syncthreads();
    if(lds[(threadIdx.x * 37) % n] != 7.0) atomicAdd(out + 3, 1.0);
}
static void report(char const* name, double* d, int wgs, int n_alloc, int n_use)
{
    double h[4];
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::printf("%s: %4d workgroups, %d doubles allocated, %d used: inside intact %.0f of %d, outside intact %.0f zero %.0f of %d, static damaged %.0f (%s)\n", name, wgs, n_alloc,
                n_use, h[0], wgs * n_alloc, h[1], h[2], wgs * (n_use - n_alloc), h[3], hipGetErrorString(hipGetLastError()));
}
int main()
{
    double* d;
    (void)hipMalloc(&d, 4 * sizeof(double));
    int const n_alloc = 5074, n_use = 8076, wgs = 1024;
    (void)hipFuncSetAttribute((void const*)probe_a, hipFuncAttributeMaxDynamicSharedMemorySize, n_alloc * 8);
    (void)hipFuncSetAttribute((void const*)probe_b, hipFuncAttributeMaxDynamicSharedMemorySize, n_alloc * 8);
    (void)hipFuncSetAttribute((void const*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipMemset(d, 0, 4 * sizeof(double));
    hipLaunchKernelGGL(probe_a, dim3(wgs), dim3(256), n_alloc * 8, 0, d, n_alloc, n_use);
    report("A first probe's shape      ", d, wgs, n_alloc, n_use);
    (void)hipMemset(d, 0, 4 * sizeof(double));
    hipLaunchKernelGGL(probe_b, dim3(1, wgs), dim3(256), n_alloc * 8, 0, d, n_alloc, n_use);
    report("B production kernel's shape", d, wgs, n_alloc, n_use);
    (void)hipMemset(d, 0, 4 * sizeof(double));
    hipLaunchKernelGGL(hog, dim3(512), dim3(256), 80 * 1024, 0, d, 10000);
    hipLaunchKernelGGL(probe_b, dim3(1, wgs), dim3(256), n_alloc * 8, 0, d, n_alloc, n_use);
    report("C after an 80 KB kernel    ", d, wgs, n_alloc, n_use);
    (void)hipFuncSetAttribute((void const*)probe_b, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipMemset(d, 0, 4 * sizeof(double));
    hipLaunchKernelGGL(probe_b, dim3(1, wgs), dim3(256), n_alloc * 8, 0, d, n_alloc, n_use);
    report("D attribute raised to 80 KB", d, wgs, n_alloc, n_use);
    return 0;
}
