// tests/emu/hip_shim -- TEST INFRASTRUCTURE ONLY.
// A host stand-in for the handful of HIP runtime calls pe_engine.cpp makes, so that the engine's host logic and
// the team-generic front code (pe_front.hpp) can be exercised with a ONE-THREAD team in a container without a GPU
// (indexing / call-order checks before any kernel is launched on real hardware).  The product library
// (phy-engine_amd/libpe_hip.so) is never built against this header and has no CPU path.
#pragma once
#include <chrono>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInsufficientDriver = 35, hipErrorNotReady = 600, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101 };
typedef struct emu_stream* hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };

inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
// (PE_EMU_DEVICES: how many 'devices' the emulation reports -- the multi-device sweep entry point is tested over two of them)
inline hipError_t hipGetDeviceCount(int* n) { char const* v = std::getenv("PE_EMU_DEVICES"); *n = v && *v ? std::atoi(v) : 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t b) { *p = std::malloc(b ? b : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
constexpr unsigned hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000;
inline hipError_t hipHostMalloc(void** p, size_t b, unsigned) { *p = std::malloc(b ? b : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t b) { std::memset(p, v, b); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t b, hipStream_t) { std::memset(p, v, b); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t b, hipMemcpyKind) { std::memcpy(d, s, b); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t b, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, b); return hipSuccess; }
inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{
    for(size_t r = 0; r < h; ++r) std::memcpy(static_cast<char*>(d) + r * dp, static_cast<const char*>(s) + r * sp, w);
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 160 * 1024; return hipSuccess; }
