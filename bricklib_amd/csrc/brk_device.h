// brk_device.h -- device-side helpers shared by the kernel translation units of libbrk.so.
#ifndef BRK_DEVICE_H
#define BRK_DEVICE_H
#include <hip/hip_runtime.h>

#include <cstdint>

#include "brick-device.h"
#include "brk_internal.h"
#include "brk_sbase.h"

namespace brklib {

#ifdef BRK_DEV_KNOBS
// development builds only (make DEV=1): unused dynamic LDS added to every stencil launch -> fewer workgroups resident per CU
// (tools/kbench -L), and which timing experiment the diagnostic kernels run (-DBRK_DIAG, tools/kbench -D)
extern size_t g_pad_lds;
extern int g_diag;
#else
constexpr size_t g_pad_lds = 0; // release launches carry no dynamic LDS
constexpr int g_diag = 0;
#endif

constexpr int kWavesPerWG = 4;
constexpr int kWG = 64 * kWavesPerWG;

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return brk_fail(BRK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Launchers of the specialised 7-point kernels (brk_star7.hip).  variant: 2 = one brick per wave, every face gathered;
// 5 / 6 = workgroups of 8 / 4 waves sharing i-faces through LDS.
int launch_star7(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step,
                 const uint32_t *ids, uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st, const uint32_t *faces);

int launch_star7_box(int stencil_id, const uint32_t *adj, const double *in, double *out, size_t step, uint32_t base,
                     const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], const Coeffs &cf, hipStream_t st,
                     const uint32_t *regmask, const uint32_t *faces);

// 13-/25-point kernels with workgroup-shared i-halos (brk_starwg.hip)
int launch_star_wg(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st, const uint32_t *faces, bool kpaired);

// 27-/125-point kernels: dense conflict-free tiles with sibling/fallback slots (brk_cubedense.hip), and the earlier padded
// tiles with shared i-halos (brk_cubewg_variants.hip)
// `cubereg`: the box entry point over a registered table -- one bit per brick "all 26 neighbours are the ids the lexicographic
// numbering with strides (s1, s2) predicts" (k_regular_mask, cube form): such a brick's adjacency row is computed, not loaded
struct CubeRegular {
  const uint32_t *mask = nullptr;
  uint32_t s1 = 0, s2 = 0;
  const uint32_t *faces = nullptr; // compact face rows of a registered table (k_face_table): with them and the link bits every wave of a
                                   // group works out the group's links from scalars -- what the LDS-DMA staged kernel k_cube_dma needs
};
int launch_cube_dense(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                      uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st, const uint32_t *linkmask, const CubeRegular &cubereg = CubeRegular());
// the 125-point kernel at four waves per SIMD for lists whose groups of four are fully linked i-neighbours of regular bricks
// (brk_cube125q.hip; the box entry point proves that with all_regular over the cube-regular mask of a registered table)
int launch_cube125_q(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, uint32_t n, const Coeffs &cf, hipStream_t st,
                     uint32_t s1, uint32_t s2, bool predicted);
int all_regular(const uint32_t *ids, uint32_t n, const uint32_t *mask, hipStream_t st, bool *out);
int launch_cube_wg(int stencil_id, int nw, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st);

} // namespace brklib
#endif
