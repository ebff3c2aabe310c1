// brk_device.h -- device-side helpers shared by the kernel translation units of libbrk.so.
#ifndef BRK_DEVICE_H
#define BRK_DEVICE_H
#include <hip/hip_runtime.h>

#include <cstdint>

#include "brick-device.h"
#include "brk_internal.h"

namespace brk {

constexpr int kWavesPerWG = 4;
constexpr int kWG = 64 * kWavesPerWG;

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return brk_fail(BRK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Launchers of the specialised 7-point kernels (brk_star7.hip).  variant: 2 = one brick per wave,
// 3 = persistent waves with register prefetch of the next brick.
int launch_star7(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step,
                 const uint32_t *ids, uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st);

int launch_star7_box(int stencil_id, const uint32_t *adj, const double *in, double *out, size_t step, uint32_t base,
                     const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], const Coeffs &cf, hipStream_t st);

// 13-/25-point kernels with workgroup-shared i-halos (brk_starwg.hip)
int launch_star_wg(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st);

// 27-/125-point kernels with workgroup-shared i-halos (brk_cubewg.hip)
int launch_cube_dense(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                      uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st);
int launch_cube_wg(int stencil_id, int nw, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st);

} // namespace brk
#endif
