// developer probe: what does gfx950 do with LDS accesses beyond a workgroup's allocation?  (round 3: a run of top levels was launched
// with 40 KB of dynamic LDS while its fronts used 64 KB -- and computed the right answers.)   hipcc --offload-arch=gfx950 -o probe lds_oob_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double lds[];
__global__ void probe(double* out, int n_alloc, int n_use)
{
    // every workgroup fills [0, n_use) with its own tag, waits, and counts how many cells still hold it
    double const tag = 1000.0 * (blockIdx.x + 1);
    for(int i = threadIdx.x; i < n_use; i += blockDim.x) lds[i] = tag + i;
    __syncthreads();
    for(volatile int spin = 0; spin < 20000; ++spin) {}
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
int main()
{
    double* d;
    hipMalloc(&d, 3 * sizeof(double));
    int const n_alloc = 5074, n_use = 8076;
    hipFuncSetAttribute((void const*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, n_alloc * 8);
    for(int wgs: {1, 1024})
    {
        hipMemset(d, 0, 3 * sizeof(double));
        hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), n_alloc * 8, 0, d, n_alloc, n_use);
        double h[3];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        std::printf("%4d workgroups, %d doubles allocated, %d used: inside intact %.0f of %d, outside intact %.0f zero %.0f of %d  (%s)\n", wgs, n_alloc, n_use, h[0],
                    wgs * n_alloc, h[1], h[2], wgs * (n_use - n_alloc), hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
