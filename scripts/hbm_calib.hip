// scripts/hbm_calib.hip -- developer tool (not part of the product): on-box HBM ceiling + FETCH_SIZE / WRITE_SIZE calibration
// for THIS engine's access shapes (SURVEY.md 8d "measure the achievable ceiling on the box with a device-to-device stream
// kernel"; MI355X_MICROARCH.md "Other access widths are uncalibrated: calibrate on a known byte count").
//
//   k_copy16    device-to-device stream copy, 16 B per lane                (ceiling; bytes = 2 x N)
//   k_read8     coalesced 8 B per lane streaming read (fp64 vectors)       (FETCH_SIZE factor for 8-B lanes)
//   k_seg128    8 B per lane, quarter-wave = one 128-byte segment at a pseudo-random position: the shape of the factor
//               kernel's update-matrix tile loads (16 consecutive rows of a column-major fp64 block)
//   k_write8    coalesced 8 B per lane streaming store                     (WRITE_SIZE factor)
//   k_wseg128   the tile-store shape: quarter-wave = one 128-byte segment
// Every kernel moves a byte count printed on stdout; run once bare (timings) and once under
// `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divide.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_calib scripts/hbm_calib.hip && /tmp/hbm_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if(e_ != hipSuccess)                                                                    \
        {                                                                                       \
            std::fprintf(stderr, "HIP error %s at %s\n", hipGetErrorString(e_), #x);            \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while(0)

using v4f = __attribute__((ext_vector_type(4))) float;

__global__ void __launch_bounds__(256) k_copy16(v4f const* __restrict__ src, v4f* __restrict__ dst, size_t n)
{
    size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for(size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_read8(double const* __restrict__ src, double* __restrict__ sink, size_t n)
{
    size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    double acc = 0.0;
    for(size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += 4 * stride)
    {
        double v[4];
#pragma unroll
        for(int q = 0; q < 4; ++q) v[q] = i + q * stride < n ? src[i + q * stride] : 0.0;
#pragma unroll
        for(int q = 0; q < 4; ++q) acc += v[q];
    }
    if(acc == 1.2345e300) sink[0] = acc;  // never true: keeps the loads
}

// segment s (16 doubles = 128 B) is read by one quarter-wave; the segment order is a fixed odd-multiplier permutation
__global__ void __launch_bounds__(256) k_seg128(double const* __restrict__ src, double* __restrict__ sink, size_t nseg)
{
    size_t const q0 = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4, nq = (static_cast<size_t>(gridDim.x) * blockDim.x) >> 4;
    int const l = threadIdx.x & 15;
    double acc = 0.0;
    for(size_t s = q0; s < nseg; s += 4 * nq)
    {
        double v[4];
#pragma unroll
        for(int q = 0; q < 4; ++q)
        {
            size_t const ss = s + q * nq < nseg ? s + q * nq : s;
            size_t const perm = (ss * 0x9E3779B1ull) % nseg;  // nseg is a power of two here: an odd multiplier permutes
            v[q] = src[perm * 16 + l];
        }
#pragma unroll
        for(int q = 0; q < 4; ++q) acc += v[q];
    }
    if(acc == 1.2345e300) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_write8(double* __restrict__ dst, size_t n)
{
    size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for(size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = static_cast<double>(i);
}

__global__ void __launch_bounds__(256) k_wseg128(double* __restrict__ dst, size_t nseg)
{
    size_t const q0 = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4, nq = (static_cast<size_t>(gridDim.x) * blockDim.x) >> 4;
    int const l = threadIdx.x & 15;
    for(size_t s = q0; s < nseg; s += nq)
    {
        size_t const perm = (s * 0x9E3779B1ull) % nseg;
        dst[perm * 16 + l] = static_cast<double>(s);
    }
}

template <class F>
static double time_ms(F&& launch, int reps)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    launch();  // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, nullptr));
    for(int r = 0; r < reps; ++r) launch();
    CHK(hipEventRecord(e1, nullptr));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipEventDestroy(e0));
    CHK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    size_t const bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096ull) << 20;  // MiB per buffer, default 4 GiB
    int const reps = argc > 2 ? std::atoi(argv[2]) : 5;
    void *a = nullptr, *b = nullptr;
    CHK(hipMalloc(&a, bytes));
    CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes));
    CHK(hipMemset(b, 0, bytes));
    int const grid = 256 * 16;
    size_t const n16 = bytes / 16, n8 = bytes / 8, nseg = bytes / 128;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, nullptr, static_cast<v4f const*>(a), static_cast<v4f*>(b), n16); }, reps);
    std::printf("{\"kernel\": \"k_copy16\", \"bytes_read\": %zu, \"bytes_written\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", bytes, bytes, ms, 2.0 * bytes / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read8, dim3(grid), dim3(256), 0, nullptr, static_cast<double const*>(a), static_cast<double*>(b), n8); }, reps);
    std::printf("{\"kernel\": \"k_read8\", \"bytes_read\": %zu, \"bytes_written\": 0, \"ms\": %.4f, \"GBps\": %.1f}\n", bytes, ms, 1.0 * bytes / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_seg128, dim3(grid), dim3(256), 0, nullptr, static_cast<double const*>(a), static_cast<double*>(b), nseg); }, reps);
    std::printf("{\"kernel\": \"k_seg128\", \"bytes_read\": %zu, \"bytes_written\": 0, \"ms\": %.4f, \"GBps\": %.1f}\n", bytes, ms, 1.0 * bytes / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write8, dim3(grid), dim3(256), 0, nullptr, static_cast<double*>(b), n8); }, reps);
    std::printf("{\"kernel\": \"k_write8\", \"bytes_read\": 0, \"bytes_written\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", bytes, ms, 1.0 * bytes / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_wseg128, dim3(grid), dim3(256), 0, nullptr, static_cast<double*>(b), nseg); }, reps);
    std::printf("{\"kernel\": \"k_wseg128\", \"bytes_read\": 0, \"bytes_written\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", bytes, ms, 1.0 * bytes / ms * 1e-6);
    std::printf("{\"launches_per_kernel\": %d}\n", reps + 1);
    CHK(hipFree(a));
    CHK(hipFree(b));
    return 0;
}
