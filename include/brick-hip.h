// brick-hip.h -- the vendor-neutral gpu* names that brick-gpu.h and the drivers use, bound to the HIP runtime.
// The reference binds them with one macro per name (include/brick-hip.h:12-26); here they are typed inline functions and
// constants, so a wrong argument is a compile error at the call site instead of inside a macro expansion.  Also pulls in
// the device side of the stencils (brick-device.h), which takes the place of dev_shl.h + generated code.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

using gpuMemcpyKind = hipMemcpyKind;
constexpr hipError_t gpuSuccess = hipSuccess;
constexpr gpuMemcpyKind gpuMemcpyHostToDevice = hipMemcpyHostToDevice;
constexpr gpuMemcpyKind gpuMemcpyDeviceToHost = hipMemcpyDeviceToHost;
constexpr hipFuncCache_t gpuFuncCachePreferL1 = hipFuncCachePreferL1;

template <class T>
inline hipError_t gpuMalloc(T **ptr, std::size_t bytes) {
  return hipMalloc(reinterpret_cast<void **>(ptr), bytes);
}
inline hipError_t gpuFree(void *ptr) { return hipFree(ptr); }
inline hipError_t gpuMemcpy(void *dst, const void *src, std::size_t bytes, gpuMemcpyKind kind) {
  return hipMemcpy(dst, src, bytes, kind);
}
inline const char *gpuGetErrorString(hipError_t e) { return hipGetErrorString(e); }
inline hipError_t gpuDeviceSynchronize() { return hipDeviceSynchronize(); }
inline hipError_t gpuDeviceSetCacheConfig(hipFuncCache_t c) { return hipDeviceSetCacheConfig(c); }

// these two take a symbol / a kernel name, so they stay macros
#define gpuMemcpyToSymbol(symbol, src, bytes) hipMemcpyToSymbol(HIP_SYMBOL(symbol), src, bytes, 0, hipMemcpyHostToDevice)
#define gpuExecKernel(kernel, blocks, threads, ...) hipLaunchKernelGGL(kernel, blocks, threads, 0, 0, __VA_ARGS__)

#include "brick-gpu.h"
#if defined(__HIP__)
#include "brick-device.h"
#include "dev_shl.h"
#endif
