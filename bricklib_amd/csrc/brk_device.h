// brk_device.h -- device-side helpers shared by the kernel translation units of libbrk.so.
#ifndef BRK_DEVICE_H
#define BRK_DEVICE_H
#include <hip/hip_runtime.h>

#include <cstdint>

#include "brk_internal.h"

namespace brk {

constexpr int kWavesPerWG = 4;
constexpr int kWG = 64 * kWavesPerWG;

struct Coeffs {
  double c[32];
};

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return brk_fail(BRK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// adjacency slot of direction (dk,dj,di), include/brick.h:237-245
__host__ __device__ constexpr int adj_slot(int dk, int dj, int di) { return (dk + 1) * 9 + (dj + 1) * 3 + (di + 1); }

// Wave-local LDS ordering.  The hardware runs one wave's DS instructions in order; these only stop the
// compiler from moving a lane's reads above other lanes' writes.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Launchers of the specialised 7-point kernels (brk_star7.hip).  variant: 2 = one brick per wave,
// 3 = persistent waves with register prefetch of the next brick.
int launch_star7(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step,
                 const uint32_t *ids, uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st);

} // namespace brk
#endif
