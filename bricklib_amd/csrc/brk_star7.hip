// brk_star7.hip -- the headline kernel: 7-point fp64 sweep over 8x8x8 bricks on gfx950.
//
// One wave64 per brick, lane = (j,i), k walked in registers.  Per brick a wave moves 4 KiB in, 4 KiB out and
// ~3 KiB of neighbour faces (which should come from L2/Infinity Cache, not HBM).  The kernel is latency/occupancy
// bound, not ALU bound (56 fp64 FMAs per lane per brick), so everything here is about bytes in flight:
//   * LDS tile of 6 KiB per wave (24 waves/CU): per k-plane a dense 10x8 block (j-halo row, 8 centre rows,
//     j-halo row; one 32-lane half reads 32 consecutive doubles = conflict-free) followed by the 8+8 i-halo values;
//     lanes on an i edge simply point their ds_read at the halo slot, so every plane costs four ds_read_b64.
//   * default kernel k_s7_wg: a 512-thread workgroup takes 8 consecutive list entries and the i-faces are read from
//     the sibling waves' tiles (the two i-face gathers were the dominant cost of the one-wave-per-brick version).
// Tried and measured without benefit on MI355X, then removed (numbers in profiles/EXPERIMENTS.md): persistent waves with a 3-deep
// register prefetch pipeline, 16 vs 24 waves/CU, 16-B loads, k-stacked bricks carrying k-faces in registers.  Earlier generations
// and the two-bricks-per-wave experiments live in brk_star7_variants.hip (`make VARIANTS=1` builds only); this file is what ships.
#include "brk_device.h"
#include "brk_star7_kernel.h"

namespace brklib {
namespace {

#ifdef BRK_VARIANTS
#define BRK_VARIANTS_PART 1
#include "brk_star7_variants.hip"
#undef BRK_VARIANTS_PART
#endif

} // namespace

int launch_star7(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step,
                 const uint32_t *ids, uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st, const uint32_t *faces) {
  const int kind = stencil_id == BRK_MPI7 ? kMpi : (stencil_id == BRK_COND7 ? kCond : kCoeff);
  // workgroups of 8 waves sharing i-faces (variant 5, the default); -DBRK_VARIANTS builds: 6 = 4-wave groups, 2 = one brick per
  // wave with every face gathered
#define BRK_LAUNCH_WG(M, W, I)                                                                                                                      \
  do {                                                                                                                                             \
    brki_note_kernel("k_s7_wg<" #M "," #W "," #I ">");                                                                                              \
    hipLaunchKernelGGL((k_s7_wg<M, W, I>), dim3(wgs), dim3(64 * W), g_pad_lds, st, adj, in, out, step, ids, first, n, Box{}, cf, nullptr, faces); \
  } while (0)
#define BRK_LAUNCH_WG_K(K, W)                      \
  do {                                            \
    if (ids) BRK_LAUNCH_WG(K, W, kModeList);      \
    else BRK_LAUNCH_WG(K, W, kModeRange);         \
  } while (0)
#define BRK_LAUNCH_WG_W(W)                        \
  do {                                            \
    if (kind == kMpi) BRK_LAUNCH_WG_K(kMpi, W);   \
    else if (kind == kCond) BRK_LAUNCH_WG_K(kCond, W); \
    else BRK_LAUNCH_WG_K(kCoeff, W);              \
  } while (0)
#ifdef BRK_VARIANTS
  bool handled;
#define BRK_VARIANTS_PART 2
#include "brk_star7_variants.hip"
#undef BRK_VARIANTS_PART
  if (!handled)
#endif
  {
    (void)variant;
    const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
    BRK_LAUNCH_WG_W(8);
  }
#undef BRK_LAUNCH_WG_W
#undef BRK_LAUNCH_WG_K
#undef BRK_LAUNCH_WG
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

// Box mode (brk_stencil_apply_box): no id list; see struct Box.
int launch_star7_box(int stencil_id, const uint32_t *adj, const double *in, double *out, size_t step, uint32_t base,
                     const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], const Coeffs &cf, hipStream_t st,
                     const uint32_t *regmask, const uint32_t *faces) {
  Box bx{base, tot[0], tot[1], {lo[0], lo[1], lo[2]}, {cnt[0], cnt[1], cnt[2]}};
  const uint32_t G = (cnt[0] + 7) / 8;
  uint32_t per_xcd = 0;
  for (uint32_t x = 0; x < 8; ++x) {
    const uint32_t rows = (cnt[1] * (x + 1)) / 8 - (cnt[1] * x) / 8;
    per_xcd = rows * cnt[2] * G > per_xcd ? rows * cnt[2] * G : per_xcd;
  }
  if (per_xcd == 0) return BRK_OK;
  const uint32_t n = cnt[0] * cnt[1] * cnt[2];
#ifdef BRK_DIAG
  if (g_diag) {
#define BRK_DIAG_CASE(D) \
  if (g_diag == D) hipLaunchKernelGGL((k_s7_wg<kCoeff, 8, kModeBox, D>), dim3(8 * per_xcd), dim3(512), g_pad_lds, st, adj, in, out, step, nullptr, 0u, n, bx, cf, regmask, faces)
    BRK_DIAG_CASE(16); BRK_DIAG_CASE(1); BRK_DIAG_CASE(2); BRK_DIAG_CASE(3); BRK_DIAG_CASE(4); BRK_DIAG_CASE(5); BRK_DIAG_CASE(8); BRK_DIAG_CASE(12); BRK_DIAG_CASE(13);
#undef BRK_DIAG_CASE
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
#endif
  brki_note_kernel(stencil_id == BRK_MPI7 ? "k_s7_wg<kMpi,8,kModeBox>" : (stencil_id == BRK_COND7 ? "k_s7_wg<kCond,8,kModeBox>" : "k_s7_wg<kCoeff,8,kModeBox>"));
  if (stencil_id == BRK_MPI7)
    hipLaunchKernelGGL((k_s7_wg<kMpi, 8, kModeBox>), dim3(8 * per_xcd), dim3(512), g_pad_lds, st, adj, in, out, step, nullptr, 0u, n, bx, cf, regmask, faces);
  else if (stencil_id == BRK_COND7)
    hipLaunchKernelGGL((k_s7_wg<kCond, 8, kModeBox>), dim3(8 * per_xcd), dim3(512), g_pad_lds, st, adj, in, out, step, nullptr, 0u, n, bx, cf, regmask, faces);
  else
    hipLaunchKernelGGL((k_s7_wg<kCoeff, 8, kModeBox>), dim3(8 * per_xcd), dim3(512), g_pad_lds, st, adj, in, out, step, nullptr, 0u, n, bx, cf, regmask, faces);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

} // namespace brklib
