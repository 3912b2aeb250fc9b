// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Host emulation of the small HIP surface the kernels in adflow_amd/csrc use, so
// that the *same kernel sources* can be compiled with g++ and exercised by the
// CPU-only CI (`pytest -m "not gpu"`) against the oracle.  It exists to check
// kernel LOGIC without a GPU; it is never loaded by adflow_amd (the product has
// no CPU path and fails loudly without the HIP library) and is never timed.
//
// Model: one OS thread runs the blocks of a grid (OpenMP across blocks); the
// threads of a block are fibers (hostsim.cpp) scheduled round-robin, so
// __syncthreads() and the wave shuffles (which are modelled as block-convergent
// exchanges) behave deterministically.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local
#define HOSTSIM 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
struct int4 { int x, y, z, w; };
struct int2 { int x, y; };
struct double2 { double x, y; };
extern thread_local uint3_ threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
typedef struct hostsim_event* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; };

inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::snprintf(p->name, 256, "hostsim (CPU emulation of the HIP kernels, test only)");
    std::snprintf(p->gcnArchName, 256, "host");
    p->multiProcessorCount = 0;
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e);
#define hipEventDisableTiming 2u
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t e);
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // one in-order queue on the host
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
// graph capture does not exist on the emulator: the library never asks for it under HOSTSIM (mg_graph_eligible), the types only have
// to compile
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)1 << 40; *t = (size_t)1 << 40; return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }

namespace hostsim {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
double shfl_exchange(double v, int srcLaneInBlock);   // block-convergent
}  // namespace hostsim

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hostsim::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { hostsim::syncthreads(); }
inline double atomicAdd(double* p, double v) {
    double old;
#pragma omp critical(hostsim_atomic)
    { old = *p; *p = old + v; }
    return old;
}
inline int hostsim_lane() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }
inline double __shfl(double v, int srcLane, int width = 64) {
    const int me = hostsim_lane();
    const int base = (me / width) * width;
    return hostsim::shfl_exchange(v, base + (srcLane % width));
}
inline double __shfl_up(double v, unsigned delta, int width = 64) {
    const int me = hostsim_lane();
    const int l = me % width;
    const int src = (l >= (int)delta) ? me - (int)delta : me;
    return hostsim::shfl_exchange(v, src);
}
inline double __shfl_down(double v, unsigned delta, int width = 64) {
    const int me = hostsim_lane();
    const int l = me % width;
    const int src = (l + (int)delta < width) ? me + (int)delta : me;
    return hostsim::shfl_exchange(v, src);
}
inline int __double2hiint(double v) { long long b; std::memcpy(&b, &v, 8); return (int)(b >> 32); }
inline int __shfl_up(int v, unsigned delta, int width = 64) { return (int)__shfl_up((double)v, delta, width); }
inline int __shfl_down(int v, unsigned delta, int width = 64) { return (int)__shfl_down((double)v, delta, width); }
inline double __shfl_xor(double v, int mask, int width = 64) {
    const int me = hostsim_lane();
    const int base = (me / width) * width;
    return hostsim::shfl_exchange(v, base + ((me % width) ^ mask));
}
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
