// scripts/copy_sweep.hip -- developer tool (GPU box): which device-to-device copy shape reaches the guide's 6.29 TB/s float4 copy
// (MI355X_MICROARCH.md:36,296)?  Prints GB/s (read + write bytes over the HIP-event time) per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using v4f = __attribute__((ext_vector_type(4))) float;
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s\n", hipGetErrorString(e_), #x); std::exit(1); } } while(0)

// grid-stride, UN loads in flight per thread
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_gs(v4f const* __restrict__ src, v4f* __restrict__ dst, size_t n)
{
    size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for(; i + (UN - 1) * stride < n; i += UN * stride)
    {
        v4f v[UN];
#pragma unroll
        for(int q = 0; q < UN; ++q) v[q] = NT ? __builtin_nontemporal_load(src + i + q * stride) : src[i + q * stride];
#pragma unroll
        for(int q = 0; q < UN; ++q)
        {
            if(NT) __builtin_nontemporal_store(v[q], dst + i + q * stride);
            else
                dst[i + q * stride] = v[q];
        }
    }
    for(; i < n; i += stride) dst[i] = src[i];
}
// each workgroup owns a contiguous chunk; a thread's UN loads are blockDim apart inside it
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_chunk(v4f const* __restrict__ src, v4f* __restrict__ dst, size_t n)
{
    size_t const per = (n + gridDim.x - 1) / gridDim.x;
    size_t const lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for(size_t base = lo; base < hi; base += static_cast<size_t>(UN) * blockDim.x)
    {
        v4f v[UN];
#pragma unroll
        for(int q = 0; q < UN; ++q)
        {
            size_t const i = base + q * blockDim.x + threadIdx.x;
            if(i < hi) v[q] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for(int q = 0; q < UN; ++q)
        {
            size_t const i = base + q * blockDim.x + threadIdx.x;
            if(i < hi)
            {
                if(NT) __builtin_nontemporal_store(v[q], dst + i);
                else
                    dst[i] = v[q];
            }
        }
    }
}
template <class F>
static void run(char const* name, F&& launch, size_t bytes, hipEvent_t e0, hipEvent_t e1)
{
    launch();
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for(int rep = 0; rep < 5; ++rep)
    {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::printf("%-28s %8.1f GB/s (%.3f ms)\n", name, 2.0 * bytes / best / 1e6, best);
}
int main()
{
    size_t const bytes = size_t(2) << 30, n = bytes / 16;
    v4f *a, *b;
    CHK(hipMalloc(&a, bytes));
    CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    run("hipMemcpyDtoD", [&] { CHK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, bytes, e0, e1);
    int const grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64};
    for(int g: grids)
    {
        char nm[64];
        std::snprintf(nm, sizeof nm, "gs<4> grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL((k_gs<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); }, bytes, e0, e1);
        std::snprintf(nm, sizeof nm, "gs<8> grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL((k_gs<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); }, bytes, e0, e1);
        std::snprintf(nm, sizeof nm, "gs<8,nt> grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL((k_gs<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); }, bytes, e0, e1);
        std::snprintf(nm, sizeof nm, "chunk<8> grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL((k_chunk<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); }, bytes, e0, e1);
        std::snprintf(nm, sizeof nm, "chunk<8,nt> grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL((k_chunk<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); }, bytes, e0, e1);
    }
    return 0;
}
