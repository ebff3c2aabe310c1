// brick-hip.h -- HIP names for the gpu* aliases used by brick-gpu.h and the drivers (reference include/brick-hip.h:12-26),
// plus the device side of the stencils (brick-device.h) which replaces dev_shl.h + generated code.
#ifndef BRICK_BRICK_HIP_H
#define BRICK_BRICK_HIP_H

#include <hip/hip_runtime.h>

#define gpuMalloc(p, s) hipMalloc(p, s)
#define gpuMemcpy(d, p, s, k) hipMemcpy(d, p, s, k)
#define gpuFree(p) hipFree(p)
#define gpuGetErrorString(e) hipGetErrorString(e)
#define gpuDeviceSynchronize() hipDeviceSynchronize()
#define gpuMemcpyToSymbol(p, d, s) hipMemcpyToSymbol(p, d, s, 0, hipMemcpyHostToDevice)
#define gpuDeviceSetCacheConfig(c) hipDeviceSetCacheConfig(c)

#define gpuFuncCachePreferL1 hipFuncCachePreferL1
#define gpuSuccess hipSuccess
#define gpuMemcpyKind hipMemcpyKind
#define gpuMemcpyHostToDevice hipMemcpyHostToDevice
#define gpuMemcpyDeviceToHost hipMemcpyDeviceToHost

#define gpuExecKernel(f, b, t, a...) hipLaunchKernelGGL(f, b, t, 0, 0, a)

#include "brick-gpu.h"
#if defined(__HIP__)
#include "brick-device.h"
#endif

#endif // BRICK_BRICK_HIP_H
