// brk_cubedense.hip -- cube stencils (27- and 125-point) on dense, bank-conflict-free LDS tiles.
//
// The padded (8+2R)^3 tile of brk_cubewg_variants.hip has a row pitch of 10/12 doubles: every ds_read_b64 of the accumulation
// loop is 2-way bank conflicted (a 32-lane half touches rows 4 pitches apart; measured: 50 % of the LDS cycles of the
// 125-point kernel are conflict cycles, profiles/r01_c_pmc_mpi125pt_512.txt) and the tile costs 8 / 13.8 KiB per wave.
//
// Here a wave's tile holds only the 8 interior columns of every (k,j) row of the halo'd brick: D[T][T][8], T = 8+2R,
// row pitch 8 doubles = 64 B.  Lane (j,i) of a 32-lane half reads row (j+dj), column (i+di) mod 8: 4 rows x 8 columns
// = 32 different banks for every tap.  The columns a tap needs beyond the brick (i+di < 0 or > 7) live, at the SAME row
// offset, in the tile of the i-neighbour -- the previous / next wave of the workgroup when the brick list runs along i
// (the usual case; tile pitch is a multiple of 32 doubles, so the bank pattern is unchanged).  A wave without such a
// sibling (first/last of the group, list padding, irregular lists) gathers its i-halo itself and parks it in the
// columns 8-R..7 / 0..R-1 of the slot on that side: slot F in front of tile 0 (shared by the group's two ends), or the
// slot of a wave that is idle.  Two ACTIVE neighbours that are not i-neighbours would need each other's slot, so the
// runs of a group are 2-coloured and processed in two phases; regular lists never take the second phase.
// Links are exact: a pair shares only if all 9 adjacency entries on the common side agree both ways, so any BrickInfo
// is handled (irregular ones at fallback speed).
// LDS per workgroup: (NW+1) tiles = 31.3 KiB (27-pt, NW 4, 5 groups/CU) / 45 KiB + 6 KiB of output-side shift slots (125-pt,
// NW 4, 3 groups/CU) / 37.5 KiB (fused double 7-point sweep, NW 4, 4 groups/CU).
// Fetch path (round 2): everything that only goes to LDS moves in 16-byte lanes -- the brick (4 instructions), the k-, j- and
// i-faces (one instruction per face for R = 2, both sides of a face in one instruction for R = 1), the radius-2 cube's edge and
// corner cells as pairs -- so that every 128-byte line is requested by one instruction of the wave.
#include "brk_device.h"
#include "brk_cubedense_kernel.h"

namespace brklib {
namespace {

#ifdef BRK_VARIANTS
#define BRK_VARIANTS_PART 1
#include "brk_cubedense_variants.hip"
#undef BRK_VARIANTS_PART
#endif

} // namespace

// variants: 27-pt 0/6 = NW 4 (default), 5 = NW 8; 125-pt 0/12 = NW 4, 3 waves/SIMD, output-side j-shift (default), 10 = input-side
// row-pipelined plane loop (the default before), 11 = the same with 5 rows in flight (spills), 5 = fully unrolled plane loop with
// static accumulators and one read burst per plane, 9 = the same with the rolled loop over a ring, 6 = NW 4 and 7 = NW 8 with the
// software-pipelined rolled loop (2 waves/SIMD)
int launch_cube_dense(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                      uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st, const uint32_t *linkmask, const CubeRegular &cubereg_in) {
#ifdef BRK_DEV_KNOBS
  const CubeRegular cubereg = (g_diag & 256) ? CubeRegular() : cubereg_in; // development bit 256: every adjacency row is loaded (A/B of the predicted rows)
#else
  const CubeRegular &cubereg = cubereg_in;
#endif
  const uint32_t wgs4 = (((n + 3) / 4 + 7) / 8) * 8;
  const bool x2 = stencil_id == BRK_MPI7_X2 || stencil_id == BRK_7PT_X2;
#define BRK_CD(KIND, NW, OCC, PIPE, WGS, LDS)                                                                                          \
  do {                                                                                                                                 \
    brki_note_kernel("k_cube_dense<" #KIND "," #NW "," #OCC "," #PIPE ">");                                                             \
    hipLaunchKernelGGL((k_cube_dense<KIND, NW, OCC, PIPE>), dim3(WGS), dim3(64 * NW), LDS, st, adj, in, out, step, ids, first, n, cf, \
                       g_diag, linkmask, cubereg.mask, cubereg.s1, cubereg.s2);                                                        \
  } while (0)
#ifdef BRK_VARIANTS
#define BRK_VARIANTS_PART 2
#include "brk_cubedense_variants.hip"
#undef BRK_VARIANTS_PART
#endif
  (void)variant;
  if (x2) BRK_CD(kDiamond25, 4, 4, 0, wgs4, g_pad_lds); // plane loop without read-ahead, 16-byte staging (see LD16)
  else if (stencil_id == BRK_27PT_COEFF) BRK_CD(kCube27, 4, 5, 1, wgs4, g_pad_lds);
  else BRK_CD(kCube125, 4, 3, 5, wgs4, g_pad_lds); // output-side j-shift: 60 + 40 LDS reads per lane, the shift through a wave-private slot
#undef BRK_CD
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

} // namespace brklib
