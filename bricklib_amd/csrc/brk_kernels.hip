// brk_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels behind include/brk.h.
//
// Execution shape shared by all stencil kernels
//   * one wavefront owns one 8x8x8 fp64 brick (4 KiB); lane = (j,i) of an 8x8 plane, the wave walks k.
//   * a 256-thread workgroup = 4 waves = 4 consecutive entries of the brick list.  Waves never
//     synchronise with each other: every wave stages its own halo tile in a private LDS slice, so the only
//     ordering needed is wave-local (LDS ops of one wave execute in issue order).
//   * the k-direction neighbours of a lane live in its own registers (a column of 8+2R values); only the
//     in-plane (j,i) neighbours go through LDS.
//   * block -> brick-list mapping is XCD-aware: blocks b and b+8 share an XCD/L2 (round-robin dispatch), so
//     block id is remapped to give each XCD one contiguous eighth of the list; consecutive bricks (which share
//     faces) then meet in the same 4 MiB L2.
// HBM-bound by design: 16 B of compulsory traffic per lattice update, no MFMA (not a contraction).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <algorithm>

#include "brk_device.h"

using namespace brklib;

namespace {

// Which entry of the brick list this wave owns (or n if none).  XCD-aware: see file header.
__device__ __forceinline__ uint32_t wave_slot(uint32_t n) {
  const uint32_t per_xcd = gridDim.x >> 3; // grid is a multiple of 8 blocks
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t w = wg * kWavesPerWG + (threadIdx.x >> 6);
  return w < n ? w : n;
}

// ================================================================================================================
// Generic kernels: one wave per brick, 4 waves per workgroup; the per-brick bodies live in include/brick-device.h
// (shared with user kernels that call brklib::brick_apply from a rewritten brick(...) line).  Earlier generation: compiled
// only with -DBRK_VARIANTS (make VARIANTS=1), for A/B runs in tools/kbench and the variant parity tests.
// ================================================================================================================
#ifdef BRK_VARIANTS
template <int KIND>
__global__ __launch_bounds__(kWG) void k_star(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                               double *__restrict__ out, size_t step,
                                               const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                               Coeffs cf) {
  __shared__ double lds[kWavesPerWG * StarTile<StarTraits<KIND>::R>::ELEMS];
  const uint32_t slot = wave_slot(n);
  if (slot >= n) return; // wave-uniform; no workgroup barrier anywhere below
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  star_brick<KIND>(lds + (threadIdx.x >> 6) * StarTile<StarTraits<KIND>::R>::ELEMS, adj, in, out, step, b, cf,
                   threadIdx.x & 63);
}

// WPW waves per workgroup: the 125-point tile is 13.5 KiB per wave, so 4-wave workgroups fit only twice per CU (8 waves);
// single-wave workgroups reach 11 waves/CU.
template <int KIND, int WPW>
__global__ __launch_bounds__(64 * WPW) void k_cube(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                    double *__restrict__ out, size_t step,
                                                    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                                    Coeffs cf) {
  __shared__ double lds[WPW * CubeTile<CubeTraits<KIND>::R>::ELEMS];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * WPW + (threadIdx.x >> 6);
  if (slot >= n) return;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  cube_brick<KIND>(lds + (threadIdx.x >> 6) * CubeTile<CubeTraits<KIND>::R>::ELEMS, adj, in, out, step, b, cf,
                   threadIdx.x & 63);
}

#endif // BRK_VARIANTS

// stencil given as data (brk_stencil_apply_taps): one wave per brick, generic staging, one LDS read + FMA per tap and plane
template <int R, int WPW>
__global__ __launch_bounds__(64 * WPW) void k_taps(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                   double *__restrict__ out, size_t step, const uint32_t *__restrict__ ids,
                                                   uint32_t first, uint32_t n, Taps tp) {
  __shared__ double lds[WPW * CubeTile<R>::ELEMS];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * WPW + (threadIdx.x >> 6);
  if (slot >= n) return;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  taps_brick<R>(lds + (threadIdx.x >> 6) * CubeTile<R>::ELEMS, adj, in, out, step, b, tp, threadIdx.x & 63);
}

// cube-shaped table of radius 3 or 4 (up to 729 taps, table in device memory): CubeWaves<R>::WAVES waves per workgroup
template <int R, int WPW>
__global__ __launch_bounds__(64 * WPW) void k_taps_far(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                       double *__restrict__ out, size_t step, const uint32_t *__restrict__ ids,
                                                       uint32_t first, uint32_t n, TapsFar tp) {
  __shared__ double lds[WPW * CubeTile<R>::ELEMS];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * WPW + (threadIdx.x >> 6);
  if (slot >= n) return;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  taps_far_brick<R>(lds + (threadIdx.x >> 6) * CubeTile<R>::ELEMS, adj, in, out, step, b, tp, threadIdx.x & 63);
}

// non-linear stencil given as a register program (brk_stencil_apply_expr): one wave per brick, generic staging
template <int R, int WPW>
__global__ __launch_bounds__(64 * WPW) void k_expr(const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out,
                                                   size_t step, const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, ExprProg pg) {
  __shared__ double lds[WPW * CubeTile<R>::ELEMS];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * WPW + (threadIdx.x >> 6);
  if (slot >= n) return;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  expr_brick<R>(lds + (threadIdx.x >> 6) * CubeTile<R>::ELEMS, adj, in, out, step, b, pg, threadIdx.x & 63);
}

// star-shaped stencil given as data (brk_stencil_apply_taps): the star staging, WPW waves per workgroup (R = 4: 27.5 KiB of LDS per wave)
template <int R, int WPW>
__global__ __launch_bounds__(64 * WPW) void k_star_taps(const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out,
                                                        size_t step, const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, StarTaps tp) {
  __shared__ double lds[WPW * StarTapTile<R>::ELEMS];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * WPW + (threadIdx.x >> 6);
  if (slot >= n) return;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  if (b == BRK_LIST_SKIP) return;
  star_taps_brick<R>(lds + (threadIdx.x >> 6) * StarTapTile<R>::ELEMS, adj, in, out, step, b, tp, threadIdx.x & 63);
}

// ================================================================================================================
// Region copies (pack / unpack / same-GPU halo links).  grid = (chunks, links); 16-B lanes, 64 KiB per block.
// ================================================================================================================
constexpr uint64_t kLinkChunk = 64 * 1024;

__global__ __launch_bounds__(256) void k_link_copy(const brk_link *__restrict__ links) {
  const brk_link L = links[blockIdx.y];
  const uint64_t beg = (uint64_t)blockIdx.x * kLinkChunk;
  if (beg >= L.bytes) return;
  const uint64_t end = beg + kLinkChunk < L.bytes ? beg + kLinkChunk : L.bytes;
  const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const char *>(L.from) + beg);
  uint4 *dst = reinterpret_cast<uint4 *>(static_cast<char *>(L.to) + beg);
  const uint32_t nvec = (uint32_t)((end - beg) >> 4);
  for (uint32_t v = threadIdx.x; v < nvec; v += 4 * 256) {
    uint4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (v + u * 256 < nvec) r[u] = src[v + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (v + u * 256 < nvec) dst[v + u * 256] = r[u];
  }
}

// ================================================================================================================
// array <-> brick and compare (include/bricksetup.h:92-221, include/brickcompare.h:30-57).
// One wave per brick of the iterated region; lane = (j,i); array rows are 64-B segments.
// ================================================================================================================
struct IterGeom {
  long sA1, sA2;       // array strides (j, k) in elements
  long sB1, sB2;       // brick-grid strides
  long pad[3];         // array-only halo
  long lo[3], cnt[3];  // brick-grid start / count per axis
};

enum { kToBrick, kToArray, kCompare };

template <int MODE>
__global__ __launch_bounds__(kWG) void k_iter(IterGeom g, double *__restrict__ arr, const uint32_t *__restrict__ grid,
                                               double *__restrict__ bdat, size_t step, double tol,
                                               unsigned long long *__restrict__ result) {
  const long nb = g.cnt[0] * g.cnt[1] * g.cnt[2];
  const long w = (long)blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
  if (w >= nb) return;
  const int lane = threadIdx.x & 63, j = lane >> 3, i = lane & 7;
  const long ti = g.lo[0] + w % g.cnt[0], tj = g.lo[1] + (w / g.cnt[0]) % g.cnt[1], tk = g.lo[2] + w / (g.cnt[0] * g.cnt[1]);
  const uint32_t b = grid[tk * g.sB2 + tj * g.sB1 + ti];
  double *bp = bdat + (size_t)b * step;
  double *ap = arr + (g.pad[2] + tk * 8) * g.sA2 + (g.pad[1] + tj * 8 + j) * g.sA1 + (g.pad[0] + ti * 8 + i);
  unsigned long long bad = 0;
  double worst = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (MODE == kToBrick)
      bp[k * 64 + lane] = ap[k * g.sA2];
    else if (MODE == kToArray)
      ap[k * g.sA2] = bp[k * 64 + lane];
    else {
      const double x = bp[k * 64 + lane], y = ap[k * g.sA2];
      const double d = fabs(x - y);
      const bool ok = (d < tol) || (d < (fabs(x) + fabs(y)) * tol); // include/brickcompare.h:36-37
      bad += ok ? 0 : 1;
      worst = d > worst ? d : worst; // NaN never compares greater: counted by `bad` instead
    }
  }
  if (MODE == kCompare) {
    if (bad) atomicAdd(&result[0], bad);
    atomicMax(&result[1], (unsigned long long)__double_as_longlong(worst));
  }
}

#ifdef BRK_DEV_KNOBS
// experiment knob (brk_set_variant, development builds only): read once per launch; atomic so that a concurrent setter is not a
// data race, but a process that changes variants while other threads launch gets either kernel
std::atomic<int> g_variant[BRK_NSTENCIL];
#endif
constexpr int kDefaultStar7Variant = 5; // 8-wave workgroups sharing i-faces through LDS (brk_star7.hip)

// defaults: stencils/fake.h:11-33
const double kDefault[BRK_NSTENCIL][10] = {{0}, {0}, {0.4, 0.1}, {0.4, 0.07, 0.03}, {0.1, 0.06, 0.045, 0.03, 0.015},
                                           {0.1, 0.04, 0.03, 0.01, 0.006, 0.004, 0.005, 0.002, 0.003, 0.001}, {0}, {0}, {0}};

int make_coeffs(int id, const double *host, Coeffs &cf) {
  std::memset(&cf, 0, sizeof cf);
  const int nc = brk_stencil_ncoeff(id);
  if (nc < 0) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
  if (!host) {
    if (id == BRK_7PT_COEFF || id == BRK_27PT_COEFF || id == BRK_COND7 || id == BRK_7PT_X2)
      return brk_fail(BRK_EINVAL, "stencil %d needs a coefficient vector", id);
    host = kDefault[id == BRK_MPI7_X2 ? BRK_MPI7 : id];
  }
  if (id == BRK_MPI7_X2 || id == BRK_7PT_X2) {
    // weights of S*S: w[dk+1][dj+1][di+1] of one sweep (7pt.py order: centre, i+1, i-1, j+1, j-1, k+1, k-1), convolved with itself
    double w[3][3][3] = {};
    if (id == BRK_MPI7_X2) {
      w[1][1][1] = host[0];
      w[1][1][2] = w[1][1][0] = w[1][2][1] = w[1][0][1] = w[2][1][1] = w[0][1][1] = host[1];
    } else {
      w[1][1][1] = host[0], w[1][1][2] = host[1], w[1][1][0] = host[2], w[1][2][1] = host[3], w[1][0][1] = host[4];
      w[2][1][1] = host[5], w[0][1][1] = host[6];
    }
    for (int t = 0; t < 25; ++t) {
      double c = 0.0;
      for (int uk = -1; uk <= 1; ++uk)
        for (int uj = -1; uj <= 1; ++uj)
          for (int ui = -1; ui <= 1; ++ui) {
            const int vk = kDiamondTaps[t].dk - uk, vj = kDiamondTaps[t].dj - uj, vi = kDiamondTaps[t].di - ui;
            if (vk < -1 || vk > 1 || vj < -1 || vj > 1 || vi < -1 || vi > 1) continue;
            c += w[uk + 1][uj + 1][ui + 1] * w[vk + 1][vj + 1][vi + 1];
          }
      cf.c[t] = c;
    }
    return BRK_OK;
  }
  if (id == BRK_27PT_COEFF) {
    // weights per offset slot; the reference applies coeff[16] to (k+1,j+1,i) instead of (k,j+1,i)
    // (stencils/3axis.cpp:195,:225), so slot (0,+1,0) gets nothing and (+1,+1,0) gets coeff[16]+coeff[25].
    for (int s = 0; s < 27; ++s) cf.c[s] = host[s];
    cf.c[adj_slot(0, 1, 0)] = 0.0;
    cf.c[adj_slot(1, 1, 0)] = host[16] + host[25];
  } else {
    for (int s = 0; s < nc; ++s) cf.c[s] = host[s];
  }
  return BRK_OK;
}

// registered adjacency tables (defined further down).  Lifetime rule: a launch path takes g_adj_mutex BEFORE it looks a derived
// table up and keeps it until the kernel that reads the table has been enqueued; brk_adjacency_release takes the same mutex and
// then frees with hipFree, which waits for enqueued work -- so a release from another thread can never free a table between
// the lookup and the launch.  The *_locked functions expect the caller to hold the mutex.
std::mutex g_adj_mutex;
// per brick, is it linked to its -i / +i neighbour (see k_link_mask)?
int link_mask_locked(const uint32_t *adj, const uint32_t **out);
// ... and the six face neighbours of every brick as one aligned 32-byte row (see k_face_table)
int face_table_locked(const uint32_t *adj, const uint32_t **out);
// ... and, per pair of id strides, one bit per brick "the neighbours are the ids a lexicographic numbering with these strides predicts"
int regular_mask_locked(const uint32_t *adj, uint32_t s1, uint32_t s2, int cube, hipStream_t st, const uint32_t **out, uint64_t *uid = nullptr);

// box_strides: the launch comes from the box entry point -- ids are base + (k * tot[1] + j) * tot[0] + i, strides {tot[0], tot[0] * tot[1]}
int launch_stencil(int id, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const double *coeff_host, brk_stream s, bool kpaired = false, bool reverse = false,
                   const uint32_t *box_strides = nullptr) {
  if (id < 0 || id >= BRK_NSTENCIL) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
  if (!adj || !in || !out) return brk_fail(BRK_EINVAL, "null device pointer");
  if (in == out) return brk_fail(BRK_EINVAL, "in and out must be different fields");
  if (step < 512) return brk_fail(BRK_EINVAL, "step_elems %zu < 512", step);
  // the kernels move halo rows and columns in 16-byte lanes
  if (((uintptr_t)in | (uintptr_t)out) % 16 || step % 2) return brk_fail(BRK_EINVAL, "brick fields must be 16-byte aligned (in %p, out %p, step_elems %zu)", (const void *)in, (void *)out, step);
  if (n == 0) return BRK_OK;
#ifdef BRK_DEV_KNOBS
  const int var = g_variant[id].load(std::memory_order_relaxed);
#else
  const int var = 0;
#endif
  Coeffs cf;
  if (int rc = make_coeffs(id, coeff_host, cf)) return rc;
  cf.reverse = reverse ? 1u : 0u;
  hipStream_t st = static_cast<hipStream_t>(s);
  const uint32_t *lmask = nullptr, *faces = nullptr;
  CubeRegular cubereg;
  std::lock_guard<std::mutex> adj_lock(g_adj_mutex); // held until the launch below is enqueued (see the lifetime rule above)
  if (id == BRK_27PT_COEFF || id == BRK_MPI125 || id == BRK_MPI7_X2 || id == BRK_7PT_X2) {
    if (int rc = link_mask_locked(adj, &lmask)) return rc; // cube kernels over a registered table: the rows of the list neighbours need not be read
    if (int rc = face_table_locked(adj, &cubereg.faces)) return rc;
    // measured (profiles/r04_h_predicted_rows.txt, interleaved): 27-point -0.8 ... -1.5 %, fused double 7-point pass -1 %, 125-point +1.7 %
    // (its waves then issue their requests earlier and collide more): the 125-point kernel keeps loading its rows
    if (box_strides && lmask && id != BRK_MPI125) {
      cubereg.s1 = box_strides[0], cubereg.s2 = box_strides[1];
      if (int rc = regular_mask_locked(adj, cubereg.s1, cubereg.s2, 1, st, &cubereg.mask)) return rc;
    }
  } else {
    if (int rc = face_table_locked(adj, &faces)) return rc; // star kernels over a registered table: one 32-byte row of face ids instead of a 108-byte row
  }
  // default kernels: 7-point family in brk_star7.hip, 13-/25-point in brk_starwg.hip, 27-/125-point and the fused double
  // 7-point sweep in brk_cubedense.hip.  Builds with -DBRK_VARIANTS also carry the earlier generations (brk_set_variant).
#ifdef BRK_VARIANTS
  if ((id == BRK_7PT_COEFF || id == BRK_MPI7 || id == BRK_COND7) && var != 1)
    return launch_star7(id, var == 0 ? kDefaultStar7Variant : var, adj, in, out, step, ids, first, n, cf, st, faces);
  if ((id == BRK_MPI13 || id == BRK_MPI25) && var != 1) return launch_star_wg(id, var, adj, in, out, step, ids, first, n, cf, st, faces, kpaired);
  if (id == BRK_MPI7_X2 || id == BRK_7PT_X2) return launch_cube_dense(id, var, adj, in, out, step, ids, first, n, cf, st, lmask, cubereg);
  if (id == BRK_27PT_COEFF || id == BRK_MPI125) {
    const int v = var;
    if (v == 0 || (v >= 5 && v <= 19)) return launch_cube_dense(id, v, adj, in, out, step, ids, first, n, cf, st, lmask, cubereg);
    if ((id == BRK_27PT_COEFF && v == 2) || (id == BRK_MPI125 && (v == 2 || v == 4 || v == 8)))
      return launch_cube_wg(id, id == BRK_MPI125 ? v : 8, adj, in, out, step, ids, first, n, cf, st);
  }
  const uint32_t wgs = (n + kWavesPerWG - 1) / kWavesPerWG;
  const dim3 grid(((wgs + 7) / 8) * 8), block(kWG);
  switch (id) { // variant 1: the generic one-wave kernels (3: 125-point with 4 waves per workgroup)
  case BRK_7PT_COEFF: hipLaunchKernelGGL(k_star<kStar7Coeff>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_COND7: hipLaunchKernelGGL(k_star<kStarCond7>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI7: hipLaunchKernelGGL(k_star<kStarMpi7>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI13: hipLaunchKernelGGL(k_star<kStarMpi13>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI25: hipLaunchKernelGGL(k_star<kStarMpi25>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_27PT_COEFF: hipLaunchKernelGGL((k_cube<kCube27, 4>), grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI125:
    if (var == 3)
      hipLaunchKernelGGL((k_cube<kCube125, 4>), grid, block, 0, st, adj, in, out, step, ids, first, n, cf);
    else
      hipLaunchKernelGGL((k_cube<kCube125, 1>), dim3(((n + 7) / 8) * 8), dim3(64), 0, st, adj, in, out, step, ids, first, n, cf);
    break;
  }
  HIP_TRY(hipGetLastError());
  return BRK_OK;
#else
  if (id == BRK_7PT_COEFF || id == BRK_MPI7 || id == BRK_COND7) return launch_star7(id, kDefaultStar7Variant, adj, in, out, step, ids, first, n, cf, st, faces);
  if (id == BRK_MPI13 || id == BRK_MPI25) return launch_star_wg(id, var, adj, in, out, step, ids, first, n, cf, st, faces, kpaired);
  return launch_cube_dense(id, 0, adj, in, out, step, ids, first, n, cf, st, lmask, cubereg);
#endif
}

// BRK_CUBE125_Q=0 in the environment (read once): the box entry point keeps k_cube_dense for the 125-point stencil also where the
// four-waves-per-SIMD kernel applies -- for A/B measurements (bench.py, tools/kbench); development builds: variant 18 does the same
bool cube125_q_enabled() {
  static const bool on = [] {
    const char *v = std::getenv("BRK_CUBE125_Q");
    return !(v && v[0] == '0');
  }();
#ifdef BRK_DEV_KNOBS
  if (g_variant[BRK_MPI125].load(std::memory_order_relaxed) != 0) return false; // any named variant (18 = "the default kernel, please") keeps k_cube_dense
#endif
  return on;
}

int make_geom(const long dimlist[3], const long padding[3], const long ghost[3], IterGeom &g, long &nb) {
  if (!dimlist || !padding || !ghost) return brk_fail(BRK_EINVAL, "null geometry");
  long a = 1, b = 1, sA[3], sB[3];
  for (int d = 0; d < 3; ++d) {
    if (dimlist[d] <= 0 || dimlist[d] % 8 || ghost[d] % 8 || padding[d] < 0 || ghost[d] < 0)
      return brk_fail(BRK_EINVAL, "extent/ghost on axis %d must be non-negative multiples of 8", d);
    sA[d] = a;
    sB[d] = b;
    a *= dimlist[d] + 2 * (padding[d] + ghost[d]);
    b *= (dimlist[d] + 2 * ghost[d]) / 8;
    g.pad[d] = padding[d];
    g.lo[d] = ghost[d] / 8;
    g.cnt[d] = dimlist[d] / 8;
  }
  g.sA1 = sA[1];
  g.sA2 = sA[2];
  g.sB1 = sB[1];
  g.sB2 = sB[2];
  nb = g.cnt[0] * g.cnt[1] * g.cnt[2];
  return BRK_OK;
}

} // namespace

#ifdef BRK_DEV_KNOBS
// development builds only (make DEV=1 / VARIANTS=1 / EXTRA_HIPFLAGS=-DBRK_DIAG; tools/kbench): not part of the product ABI
namespace brklib {
size_t g_pad_lds = 0;
int g_diag = 0;
}
extern "C" void brk_debug_pad_lds(size_t bytes) { brklib::g_pad_lds = bytes; }
extern "C" void brk_debug_diag(int d) { brklib::g_diag = d; }

extern "C" int brk_set_variant(int id, int v) {
  if (id < 0 || id >= BRK_NSTENCIL) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
#ifndef BRK_VARIANTS
  if (v != 0 && !((id == BRK_MPI13 || id == BRK_MPI25) && v == 3)) return brk_fail(BRK_EINVAL, "kernel variant %d: this build carries the default kernels only (make -C bricklib_amd/csrc VARIANTS=1 adds the earlier generations)", v);
#endif
  g_variant[id] = v;
  return BRK_OK;
}
#endif

// Id list of a structured box in the traversal order of bricklib_amd/order.py (per-XCD slabs of j-rows swept over all k,
// left-over rows cut along k, one skipped entry behind rows that are not a multiple of 8, XCD parts padded to equal length).
// Lists are owned by the library: a small per-process cache keyed by (device, geometry), least recently used entry replaced.
// Lifetime rules (brk.h promises thread safety for distinct streams and devices):
//   * the cache mutex is held from the lookup until the launch that reads the list has been enqueued and an event has been
//     recorded behind it on the launch stream -- another thread cannot replace the entry in between;
//   * a replaced list is not freed in place and nothing waits under the mutex: it moves to a retired list and is released by a
//     later call once hipEventQuery says its last reader has finished (no device-wide synchronisation, no hipSetDevice).
struct BoxList {
  uint32_t key[12];
  uint32_t *dev;
  uint32_t n;
  uint64_t stamp;
  int device;
  hipEvent_t last_use; // recorded behind the most recent launch that reads `dev`
  // 125-point kernel at four waves per SIMD (brk_cube125q.hip): is every brick of this list regular in the registered table whose
  // cube-regular mask has the unique id q_uid?  (one reduction per list and mask; 0 = not examined yet)
  uint64_t q_uid;
  bool q_ok;
};
constexpr int kBoxCache = 16;
BoxList g_box[kBoxCache];
uint64_t g_box_clock = 0;
std::mutex g_box_mutex;
struct RetiredList {
  uint32_t *dev;
  hipEvent_t last_use;
};
std::vector<RetiredList> g_box_retired;

// caller holds g_box_mutex: free the retired lists whose last reader has finished; never blocks
void reap_retired_locked() {
  size_t keep = 0;
  for (size_t q = 0; q < g_box_retired.size(); ++q) {
    if (hipEventQuery(g_box_retired[q].last_use) == hipSuccess) {
      (void)hipFree(g_box_retired[q].dev);
      (void)hipEventDestroy(g_box_retired[q].last_use);
    } else {
      (void)hipGetLastError(); // hipErrorNotReady is not an error of this call
      g_box_retired[keep++] = g_box_retired[q];
    }
  }
  g_box_retired.resize(keep);
}

// caller holds g_box_mutex
// rows_per_block: j-rows of an XCD slab swept over all k before the next block (the distance between the two uses of a k-face).
// kpair: the k-paired order of k_star_kp -- inside a block two k-levels at a time, every run of 4 i-neighbours at k followed by the
// same run at k+1 (rows padded to multiples of 4 entries; an odd last level is paired with skipped entries)
// icols (k-paired lists only; 0 = whole rows): the rows of a block are cut along i into chunks of icols bricks (a multiple of 4) and each
// chunk is swept over all k before the next.  A k-paired block of 8 rows x 64 bricks moves 4 MiB per pair of levels -- the whole L2 of an
// XCD -- between the request of a brick's planes as the k-halo of the pair below and their request as the brick itself; chunks of 32
// bricks halve that distance without fewer rows per block (which costs j-face sharing).  Measured for the radius-4 star at 512^3
// (profiles/r05_d_*): fabric reads 1.80 -> 1.32 GB per sweep, 0.4422 -> 0.4356 ms; the radius-2 star loses (0.3771 -> 0.3853 ms).
int box_list_locked(uint32_t base, const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], uint32_t rows_per_block, bool kpair,
                    uint32_t icols, BoxList **entry) {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  if (!kpair || icols >= cnt[0]) icols = 0;
  const uint32_t key[12] = {(uint32_t)device, base, tot[0], tot[1], tot[2], lo[0], lo[1], lo[2], cnt[0], cnt[1], cnt[2], rows_per_block | (kpair ? 0x100u : 0u) | (icols << 16)};
  int victim = 0;
  for (int q = 0; q < kBoxCache; ++q) {
    if (g_box[q].dev && std::memcmp(g_box[q].key, key, sizeof key) == 0) {
      g_box[q].stamp = ++g_box_clock;
      *entry = &g_box[q];
      return BRK_OK;
    }
    if (g_box[q].stamp < g_box[victim].stamp) victim = q;
  }
  const uint32_t ni = cnt[0], nj = cnt[1], nk = cnt[2], rowlen = ni % 8 ? ni + 1 : ni;
  auto id_at = [&](uint32_t k, uint32_t j, uint32_t i) { return base + ((lo[2] + k) * tot[1] + (lo[1] + j)) * tot[0] + lo[0] + i; };
  std::vector<std::vector<uint32_t>> part(8);
  const uint32_t jbase = nj / 8, rem = nj % 8;
  auto push_row = [&](std::vector<uint32_t> &v, uint32_t k, uint32_t j) {
    for (uint32_t i = 0; i < ni; ++i) v.push_back(id_at(k, j, i));
    if (rowlen != ni) v.push_back(BRK_LIST_SKIP);
  };
  // two k-levels of one row interleaved in runs of 4: [i..i+3 at k][i..i+3 at k+1]; k1 == nk: the upper level does not exist
  auto push_row_pair = [&](std::vector<uint32_t> &v, uint32_t k0, uint32_t k1, uint32_t j, uint32_t ia, uint32_t ib) {
    for (uint32_t i0 = ia; i0 < ib; i0 += 4)
      for (uint32_t k : {k0, k1})
        for (uint32_t i = i0; i < i0 + 4; ++i) v.push_back(i < ni && k < nk ? id_at(k, j, i) : BRK_LIST_SKIP);
  };
  const uint32_t chunk = icols ? (icols + 3) / 4 * 4 : (ni + 3) / 4 * 4; // k-paired lists: bricks of a row per chunk
  for (uint32_t x = 0; x < 8; ++x) {
    for (uint32_t j0 = x * jbase; j0 < (x + 1) * jbase; j0 += rows_per_block)
      for (uint32_t ia = 0; ia < (kpair ? ni : 1u); ia += chunk)
        for (uint32_t k = 0; k < nk; k += kpair ? 2 : 1)
          for (uint32_t j = j0; j < j0 + rows_per_block && j < (x + 1) * jbase; ++j) {
            if (kpair) push_row_pair(part[x], k, k + 1, j, ia, ia + chunk < ni ? ia + chunk : ni);
            else push_row(part[x], k, j);
          }
    if (rem) {
      const uint32_t ka = (uint32_t)((uint64_t)nk * x / 8), kb = (uint32_t)((uint64_t)nk * (x + 1) / 8);
      for (uint32_t k = ka; k < kb; k += kpair ? 2 : 1)
        for (uint32_t j = nj - rem; j < nj; ++j) {
          if (kpair) push_row_pair(part[x], k, k + 1 < kb ? k + 1 : nk, j, 0, ni);
          else push_row(part[x], k, j);
        }
    }
  }
  size_t per = 0;
  for (auto &v : part) per = v.size() > per ? v.size() : per;
  per = (per + 7) / 8 * 8;
  std::vector<uint32_t> host(per * 8, BRK_LIST_SKIP);
  for (uint32_t x = 0; x < 8; ++x) std::copy(part[x].begin(), part[x].end(), host.begin() + x * per);
  BoxList &e = g_box[victim];
  reap_retired_locked();
  if (e.dev) { // the replaced list may still be read by a launch in flight (any device): it is released by a later call
    g_box_retired.push_back({e.dev, e.last_use});
    e.dev = nullptr;
  }
  HIP_TRY(hipMalloc(&e.dev, host.size() * sizeof(uint32_t)));
  hipError_t r = hipMemcpy(e.dev, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (r == hipSuccess) r = hipEventCreateWithFlags(&e.last_use, hipEventDisableTiming);
  if (r != hipSuccess) {
    (void)hipFree(e.dev);
    e.dev = nullptr;
    HIP_TRY(r);
  }
  std::memcpy(e.key, key, sizeof key);
  e.q_uid = 0, e.q_ok = false;
  e.n = (uint32_t)host.size();
  e.device = device;
  e.stamp = ++g_box_clock;
  *entry = &e;
  return BRK_OK;
}


// ---- registered (immutable) adjacency tables -----------------------------------------------------------------------------------
// brk_adjacency_register(adj, nbricks): the caller promises that the table does not change until brk_adjacency_release(adj).  The
// library may then keep derived tables next to it.  So far one kind: per pair of id strides (s1, s2) a bit per brick saying that its
// six face neighbours are b -+ 1, b -+ s1, b -+ s2 -- what the lexicographic numbering of a box predicts -- which lets the 7-point
// box kernel skip the adjacency rows of those bricks.  Unregistered tables work exactly as before.
namespace {
// cube != 0: all 26 neighbours (and the brick's own entry) are the predicted ones -- the cube kernels' form of the same promise
__global__ void k_regular_mask(const uint32_t *__restrict__ adj, uint32_t nbricks, uint32_t s1, uint32_t s2, uint32_t *__restrict__ mask, int cube) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (b < nbricks) {
    const uint32_t *row = adj + (size_t)b * 27;
    ok = row[adj_slot(0, 0, -1)] == b - 1 && row[adj_slot(0, 0, 1)] == b + 1 && row[adj_slot(0, -1, 0)] == b - s1 &&
         row[adj_slot(0, 1, 0)] == b + s1 && row[adj_slot(-1, 0, 0)] == b - s2 && row[adj_slot(1, 0, 0)] == b + s2 &&
         b >= s2 && b + s2 < nbricks; // the predicted ids exist
    if (cube) {
      ok = ok && b >= s2 + s1 + 1 && (uint64_t)b + s2 + s1 + 1 < nbricks;
      for (int q = 0; q < 27 && ok; ++q) ok = row[q] == b + (uint32_t)(q / 9 - 1) * s2 + (uint32_t)((q / 3) % 3 - 1) * s1 + (uint32_t)(q % 3 - 1);
    }
  }
  const unsigned long long m = __ballot(ok);
  if ((threadIdx.x & 63) == 0) {
    mask[2 * (b >> 6)] = (uint32_t)m;
    mask[2 * (b >> 6) + 1] = (uint32_t)(m >> 32);
  }
}

// Two bits per brick b of a registered table, words 2(b/32) and 2(b/32)+1: is b LINKED to its -i neighbour l = adj[b](0,0,-1) / to
// its +i neighbour r = adj[b](0,0,+1) in the sense of the cube kernels (brk_cubedense.hip): the two bricks see the same world across
// their common i-face -- right(dk,dj,-1) == left(dk,dj,0) and right(dk,dj,0) == left(dk,dj,+1) for all nine (dk,dj), which includes
// right.im == left and left.ip == right.  A wave whose list neighbour IS that adjacency neighbour then shares halo columns with it
// through LDS without reading the neighbour's adjacency row.
__global__ void k_link_mask(const uint32_t *__restrict__ adj, uint32_t nbricks, uint32_t *__restrict__ mask) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  bool okl = false, okr = false;
  if (b < nbricks) {
    const uint32_t *row = adj + (size_t)b * 27;
    auto linked_pair = [&](const uint32_t *left, const uint32_t *right) {
      bool ok = true;
      for (int q = 0; q < 27; ++q)
        if (q % 3 != 2) ok = ok && right[q] == left[q + 1];
      return ok;
    };
    const uint32_t l = row[adj_slot(0, 0, -1)], r = row[adj_slot(0, 0, 1)];
    okl = l < nbricks && linked_pair(adj + (size_t)l * 27, row);
    okr = r < nbricks && linked_pair(row, adj + (size_t)r * 27);
  }
  const unsigned long long ml = __ballot(okl), mr = __ballot(okr);
  if ((threadIdx.x & 63) == 0) { // bricks 64q .. 64q+63: words 4q (left, low half), 4q+1 (right, low), 4q+2 (left, high), 4q+3 (right, high)
    uint32_t *w = mask + 4 * (size_t)(b >> 6);
    w[0] = (uint32_t)ml, w[1] = (uint32_t)mr, w[2] = (uint32_t)(ml >> 32), w[3] = (uint32_t)(mr >> 32);
  }
}

// The six face neighbours of every brick of a registered table as ONE aligned 32-byte row {k-, k+, j-, j+, i-, i+, 0, 0}: what the star
// kernels need of the 108-byte adjacency row (which straddles two or three 64-byte lines: 60 MB of reads per 512^3 sweep, against 8 MB)
__global__ void k_face_table(const uint32_t *__restrict__ adj, uint32_t nbricks, uint32_t *__restrict__ faces) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbricks) return;
  const uint32_t *row = adj + (size_t)b * 27;
  uint4 lo = {row[adj_slot(-1, 0, 0)], row[adj_slot(1, 0, 0)], row[adj_slot(0, -1, 0)], row[adj_slot(0, 1, 0)]};
  uint4 hi = {row[adj_slot(0, 0, -1)], row[adj_slot(0, 0, 1)], 0u, 0u};
  reinterpret_cast<uint4 *>(faces)[2 * (size_t)b] = lo;
  reinterpret_cast<uint4 *>(faces)[2 * (size_t)b + 1] = hi;
}

struct AdjEntry {
  const uint32_t *adj;
  uint32_t nbricks;
  int device;
  struct Mask {
    uint32_t s1, s2;
    int cube; // 0: the six face neighbours are the predicted ones (7-point box kernel); 1: all 26 (cube kernels)
    uint32_t *dev;
    uint64_t stamp;
    uint64_t uid; // never reused: what a cached verdict about this mask is keyed by (device pointers are recycled)
  };
  std::vector<Mask> masks;      // (s1, s2) -> device bit mask, at most kMaxMasks of them (least recently used one replaced)
  uint32_t *linkmask = nullptr; // k_link_mask, built by brk_adjacency_register
  uint32_t *faces = nullptr;    // k_face_table, built by brk_adjacency_register
};
constexpr size_t kMaxMasks = 8;
uint64_t g_mask_clock = 0;
std::vector<AdjEntry> g_adj;

void free_entry(AdjEntry &e) { // hipFree waits for the kernels that still read a table
  for (auto &m : e.masks) (void)hipFree(m.dev);
  if (e.linkmask) (void)hipFree(e.linkmask);
  if (e.faces) (void)hipFree(e.faces);
  e.masks.clear();
  e.linkmask = e.faces = nullptr;
}

AdjEntry *find_locked(const uint32_t *adj) {
  for (AdjEntry &e : g_adj)
    if (e.adj == adj) return &e;
  return nullptr;
}

int same_device(const AdjEntry &e) {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  if (device != e.device) return brk_fail(BRK_EINVAL, "adjacency table registered on device %d used on device %d", e.device, device);
  return BRK_OK;
}

// the face-regularity mask of a registered table for id strides (s1, s2), built on first use of that pair (that one call is
// synchronous on `st`: the strides are a property of the launch, not of the table); nullptr (and BRK_OK) when the table is not
// registered.  Caller holds g_adj_mutex.
int regular_mask_locked(const uint32_t *adj, uint32_t s1, uint32_t s2, int cube, hipStream_t st, const uint32_t **out, uint64_t *uid) {
  *out = nullptr;
  AdjEntry *e = find_locked(adj);
  if (!e) return BRK_OK;
  for (auto &m : e->masks)
    if (m.s1 == s1 && m.s2 == s2 && m.cube == cube) {
      m.stamp = ++g_mask_clock;
      *out = m.dev;
      if (uid) *uid = m.uid;
      return BRK_OK;
    }
  if (int rc = same_device(*e)) return rc;
  const uint32_t words = 2 * ((e->nbricks + 255) / 256 * 4);
  uint32_t *mask = nullptr;
  HIP_TRY(hipMalloc(&mask, (size_t)words * sizeof(uint32_t)));
  hipLaunchKernelGGL(k_regular_mask, dim3((e->nbricks + 255) / 256), dim3(256), 0, st, adj, e->nbricks, s1, s2, mask, cube);
  hipError_t r = hipGetLastError();
  if (r == hipSuccess) r = hipStreamSynchronize(st);
  if (r != hipSuccess) {
    (void)hipFree(mask);
    HIP_TRY(r);
  }
  if (e->masks.size() >= kMaxMasks) { // bounded: the least recently used pair of strides goes (hipFree waits for its readers)
    size_t victim = 0;
    for (size_t q = 1; q < e->masks.size(); ++q)
      if (e->masks[q].stamp < e->masks[victim].stamp) victim = q;
    (void)hipFree(e->masks[victim].dev);
    e->masks.erase(e->masks.begin() + victim);
  }
  const uint64_t stamp = ++g_mask_clock;
  e->masks.push_back({s1, s2, cube, mask, stamp, stamp});
  *out = mask;
  if (uid) *uid = stamp;
  return BRK_OK;
}

int face_table_locked(const uint32_t *adj, const uint32_t **out) {
  const AdjEntry *e = find_locked(adj);
  *out = e ? e->faces : nullptr;
  return e ? same_device(*e) : BRK_OK;
}

// for brk_expr_apply (brk_jit.hip): the compact face rows of a registered table, or nullptr.  Lifetime rule as for every launch path:
// g_adj_mutex is taken here and stays taken until the caller has enqueued its launch (brki_adj_unlock)
extern "C" int brki_face_table_lock(const uint32_t *adj, const uint32_t **out) {
  g_adj_mutex.lock();
  const int rc = face_table_locked(adj, out);
  if (rc != BRK_OK) g_adj_mutex.unlock();
  return rc;
}
extern "C" void brki_adj_unlock(void) { g_adj_mutex.unlock(); }

int link_mask_locked(const uint32_t *adj, const uint32_t **out) {
  const AdjEntry *e = find_locked(adj);
  *out = e ? e->linkmask : nullptr;
  return e ? same_device(*e) : BRK_OK;
}
} // namespace

// The two derived tables that depend on the adjacency alone are built HERE (setup time, null stream, synchronous), so that the
// launch paths never allocate or synchronise for them: usable under stream capture and from several streams at once.
extern "C" int brk_adjacency_register(const uint32_t *adj_dev, uint32_t nbricks) {
  if (!adj_dev || nbricks == 0) return brk_fail(BRK_EINVAL, "brk_adjacency_register: null table / no bricks");
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_adj_mutex);
  if (find_locked(adj_dev)) return brk_fail(BRK_EINVAL, "brk_adjacency_register: table %p is registered already", (const void *)adj_dev);
  AdjEntry e{adj_dev, nbricks, device, {}, nullptr, nullptr};
  const dim3 grid((nbricks + 255) / 256), block(256);
  hipError_t r = hipMalloc(&e.faces, (size_t)nbricks * 8 * sizeof(uint32_t));
  if (r == hipSuccess) r = hipMalloc(&e.linkmask, (size_t)grid.x * 16 * sizeof(uint32_t));
  if (r == hipSuccess) {
    hipLaunchKernelGGL(k_face_table, grid, block, 0, nullptr, adj_dev, nbricks, e.faces);
    hipLaunchKernelGGL(k_link_mask, grid, block, 0, nullptr, adj_dev, nbricks, e.linkmask);
    r = hipGetLastError();
  }
  if (r == hipSuccess) r = hipStreamSynchronize(nullptr);
  if (r != hipSuccess) {
    free_entry(e);
    HIP_TRY(r);
  }
  g_adj.push_back(e);
  return BRK_OK;
}

extern "C" int brk_adjacency_release(const uint32_t *adj_dev) {
  std::lock_guard<std::mutex> lock(g_adj_mutex);
  for (size_t q = 0; q < g_adj.size(); ++q)
    if (g_adj[q].adj == adj_dev) {
      free_entry(g_adj[q]);
      g_adj.erase(g_adj.begin() + q);
      return BRK_OK;
    }
  return brk_fail(BRK_EINVAL, "brk_adjacency_release: table %p is not registered", (const void *)adj_dev);
}

// Orderly end of the library's use of the GPU, for a process that is about to leave main(): every launch and copy the library has
// enqueued on the current device has finished when this returns, and what the library itself holds on a device -- the box id lists
// with their events, retired lists, the derived tables of adjacency tables that are still registered -- is released by the
// CALLING thread, not by static destructors racing the HIP runtime's own exit handlers.  Caller memory is not
// touched; the library stays usable (caches are rebuilt on demand, registrations are forgotten).
// Tap tables too large for a kernel argument (cube-shaped, radius 3 and 4): kept on the device per (device, content); the same table
// arrives again with every sweep of a time loop.  Uploaded once, synchronously; at most kMaxFarTables live (the oldest goes, after the
// device has drained).  Layout: n ints (tap_rel), padded to 8 bytes, then n doubles.
struct FarTable {
  int device = 0, n = 0, radius = 0;
  std::vector<int32_t> offsets;
  std::vector<double> weights;
  char *dev = nullptr;
};
static std::mutex g_far_mutex;
static std::vector<FarTable> g_far;
constexpr size_t kMaxFarTables = 32;

static int far_table_locked(int radius, uint32_t ntaps, const int32_t *offsets, const double *weights, TapsFar *out) {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  for (FarTable &f : g_far)
    if (f.device == device && f.n == (int)ntaps && f.radius == radius && std::memcmp(f.offsets.data(), offsets, ntaps * 12) == 0 &&
        std::memcmp(f.weights.data(), weights, ntaps * 8) == 0) {
      out->n = f.n, out->rel = reinterpret_cast<const int *>(f.dev), out->w = reinterpret_cast<const double *>(f.dev + (((size_t)f.n * 4 + 7) & ~(size_t)7));
      return BRK_OK;
    }
  if (g_far.size() >= kMaxFarTables) {
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(g_far.front().dev);
    g_far.erase(g_far.begin());
  }
  FarTable f;
  f.device = device, f.n = (int)ntaps, f.radius = radius;
  f.offsets.assign(offsets, offsets + 3 * ntaps);
  f.weights.assign(weights, weights + ntaps);
  const size_t woff = ((size_t)ntaps * 4 + 7) & ~(size_t)7;
  std::vector<char> host(woff + (size_t)ntaps * 8, 0);
  for (uint32_t t = 0; t < ntaps; ++t) {
    const int di = offsets[3 * t], dj = offsets[3 * t + 1], dk = offsets[3 * t + 2];
    reinterpret_cast<int *>(host.data())[t] = radius == 3 ? tap_rel<3>(di, dj, dk) : tap_rel<4>(di, dj, dk);
  }
  std::memcpy(host.data() + woff, weights, (size_t)ntaps * 8);
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&f.dev), host.size()));
  if (hipError_t e = hipMemcpy(f.dev, host.data(), host.size(), hipMemcpyHostToDevice); e != hipSuccess) {
    (void)hipFree(f.dev);
    return brk_fail(BRK_EHIP, "tap table upload: %s", hipGetErrorString(e));
  }
  out->n = f.n, out->rel = reinterpret_cast<const int *>(f.dev), out->w = reinterpret_cast<const double *>(f.dev + woff);
  g_far.push_back(std::move(f));
  return BRK_OK;
}

extern "C" int brk_finalize(void) {
  HIP_TRY(hipDeviceSynchronize());
  brki_jit_finalize();
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  {
    std::lock_guard<std::mutex> lock(g_box_mutex);
    auto on = [&](int d) { // lists live on the device they were built on
      if (d != device) (void)hipSetDevice(d);
    };
    for (BoxList &e : g_box)
      if (e.dev) {
        on(e.device);
        (void)hipEventSynchronize(e.last_use);
        (void)hipFree(e.dev);
        (void)hipEventDestroy(e.last_use);
        if (e.device != device) (void)hipSetDevice(device);
        e = BoxList{};
      }
    for (RetiredList &r : g_box_retired) { // the event knows its device; hipFree finds the allocation's
      (void)hipEventSynchronize(r.last_use);
      (void)hipFree(r.dev);
      (void)hipEventDestroy(r.last_use);
    }
    g_box_retired.clear();
  }
  {
    std::lock_guard<std::mutex> lock(g_adj_mutex);
    for (AdjEntry &e : g_adj) free_entry(e);
    g_adj.clear();
  }
  {
    std::lock_guard<std::mutex> lock(g_far_mutex);
    for (FarTable &f : g_far) (void)hipFree(f.dev);
    g_far.clear();
  }
  HIP_TRY(hipDeviceSynchronize());
  return BRK_OK;
}

extern "C" int brk_stencil_apply_range(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                                       uint32_t begin, uint32_t end, const double *coeff_host, brk_stream s) {
  if (end < begin) return brk_fail(BRK_EINVAL, "brick range [%u,%u) is reversed", begin, end);
  return launch_stencil(id, adj, in, out, step, nullptr, begin, end - begin, coeff_host, s);
}

extern "C" int brk_stencil_apply_list(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                                      const uint32_t *ids, uint32_t n, const double *coeff_host, brk_stream s) {
  if (n && !ids) return brk_fail(BRK_EINVAL, "null brick id list");
  return launch_stencil(id, adj, in, out, step, ids, 0, n, coeff_host, s);
}

extern "C" int brk_stencil_apply_list_kpaired(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                                              const uint32_t *ids, uint32_t n, const double *coeff_host, brk_stream s) {
  if (n && !ids) return brk_fail(BRK_EINVAL, "null brick id list");
  return launch_stencil(id, adj, in, out, step, ids, 0, n, coeff_host, s, true);
}

extern "C" int brk_stencil_apply_range_ex(int id, const uint32_t *adj, const double *in, double *out, size_t step, uint32_t begin,
                                          uint32_t end, const double *coeff_host, uint32_t flags, brk_stream s) {
  if (flags & ~(uint32_t)BRK_SWEEP_REVERSE) return brk_fail(BRK_EINVAL, "brk_stencil_apply_range_ex: flags %#x (only BRK_SWEEP_REVERSE applies to a range)", flags);
  if (end < begin) return brk_fail(BRK_EINVAL, "brick range [%u,%u) is reversed", begin, end);
  return launch_stencil(id, adj, in, out, step, nullptr, begin, end - begin, coeff_host, s, false, (flags & BRK_SWEEP_REVERSE) != 0);
}

extern "C" int brk_stencil_apply_list_ex(int id, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                                         uint32_t n, const double *coeff_host, uint32_t flags, brk_stream s) {
  if (flags & ~(uint32_t)(BRK_SWEEP_KPAIRED | BRK_SWEEP_REVERSE)) return brk_fail(BRK_EINVAL, "brk_stencil_apply_list_ex: unknown flags %#x", flags);
  if (n && !ids) return brk_fail(BRK_EINVAL, "null brick id list");
  return launch_stencil(id, adj, in, out, step, ids, 0, n, coeff_host, s, (flags & BRK_SWEEP_KPAIRED) != 0, (flags & BRK_SWEEP_REVERSE) != 0);
}

extern "C" int brk_stencil_apply_taps(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                                      uint32_t first, uint32_t n, uint32_t ntaps, const int32_t *offsets_host,
                                      const double *weights_host, brk_stream s) {
  if (!adj || !in || !out || in == out) return brk_fail(BRK_EINVAL, "bad device pointers");
  if (step < 512) return brk_fail(BRK_EINVAL, "step_elems %zu < 512", step);
  if (ntaps == 0 || ntaps > (uint32_t)kMaxFarTaps || !offsets_host || !weights_host)
    return brk_fail(BRK_EINVAL, "1..%d taps with host offset and weight arrays expected (got %u)", kMaxFarTaps, ntaps);
  if (n == 0) return BRK_OK;
  // shape of the table: taps inside the 125-point cube (|offset| <= 2 on every axis) run on the staged (8+2R)^3 tile with the table as a
  // kernel argument; star-shaped taps (one axis per tap) of radius 2..4 on the star staging; cube-shaped taps of radius 3 and 4 (up to
  // 9^3) run on the larger tiles with the table in device memory
  int maxabs = 0;
  bool star = true;
  for (uint32_t t = 0; t < ntaps; ++t) {
    int nonzero = 0;
    for (int a = 0; a < 3; ++a) {
      const int o = offsets_host[3 * t + a];
      maxabs = std::max(maxabs, o < 0 ? -o : o);
      nonzero += o != 0;
    }
    star = star && nonzero <= 1;
  }
  hipStream_t st = static_cast<hipStream_t>(s);
  if (maxabs > 4) return brk_fail(BRK_EINVAL, "taps reach offset %d: tables are limited to [-4,4] per axis", maxabs);
  if (maxabs > 2 && (!star || ntaps > (uint32_t)kMaxStarTaps)) {
    TapsFar tf;
    std::lock_guard<std::mutex> lock(g_far_mutex); // until the launch is enqueued: an eviction by another thread drains the device first, then frees
    if (int rc = far_table_locked(maxabs, ntaps, offsets_host, weights_host, &tf)) return rc;
    if (maxabs == 3) {
      brki_note_kernel("k_taps_far<3,2>");
      hipLaunchKernelGGL((k_taps_far<3, 2>), dim3((((n + 1) / 2 + 7) / 8) * 8), dim3(128), 0, st, adj, in, out, step, ids, first, n, tf);
    } else {
      brki_note_kernel("k_taps_far<4,1>");
      hipLaunchKernelGGL((k_taps_far<4, 1>), dim3(((n + 7) / 8) * 8), dim3(64), 0, st, adj, in, out, step, ids, first, n, tf);
    }
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  // star-shaped tables of radius 2 run on the star staging too (13 taps: 0.649 vs 0.705 ms per 512^3 sweep on the cube tile); radius 1
  // stays on the cube tile (0.528 vs 0.575)
  if (maxabs > 2 || (star && maxabs == 2)) {
    StarTaps sp;
    std::memset(&sp, 0, sizeof sp);
    sp.n = (int)ntaps;
    for (uint32_t t = 0; t < ntaps; ++t) {
      const int di = offsets_host[3 * t], dj = offsets_host[3 * t + 1], dk = offsets_host[3 * t + 2];
      sp.axis[t] = dk ? 2 : (dj ? 1 : 0);
      sp.off[t] = dk ? dk : (dj ? dj : di);
      sp.w[t] = weights_host[t];
    }
    if (maxabs == 2) {
      brki_note_kernel("k_star_taps<2,2>");
      hipLaunchKernelGGL((k_star_taps<2, 2>), dim3((((n + 1) / 2 + 7) / 8) * 8), dim3(128), 0, st, adj, in, out, step, ids, first, n, sp);
    } else {
      brki_note_kernel("k_star_taps<4,1>");
      hipLaunchKernelGGL((k_star_taps<4, 1>), dim3(((n + 7) / 8) * 8), dim3(64), 0, st, adj, in, out, step, ids, first, n, sp);
    }
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  if (ntaps > (uint32_t)kMaxTaps) return brk_fail(BRK_EINVAL, "%u taps inside the 125-point cube (at most %d: one per point)", ntaps, kMaxTaps);
  Taps tp;
  tp.n = (int)ntaps;
  tp.radius = maxabs > 1 ? 2 : 1;
  for (uint32_t t = 0; t < ntaps; ++t) {
    const int di = offsets_host[3 * t], dj = offsets_host[3 * t + 1], dk = offsets_host[3 * t + 2];
    tp.rel[t] = tp.radius == 1 ? tap_rel<1>(di, dj, dk) : tap_rel<2>(di, dj, dk);
    tp.w[t] = weights_host[t];
  }
  const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
  brki_note_kernel(tp.radius == 1 ? "k_taps<1,4>" : "k_taps<2,4>");
  if (tp.radius == 1) hipLaunchKernelGGL((k_taps<1, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, tp);
  else hipLaunchKernelGGL((k_taps<2, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, tp);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

extern "C" int brk_stencil_apply_expr(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, uint32_t first,
                                      uint32_t n, const brk_expr_op *ops_host, uint32_t nops, uint32_t result_reg, brk_stream s) {
  if (!adj || !in || !out || in == out) return brk_fail(BRK_EINVAL, "bad device pointers");
  if (step < 512) return brk_fail(BRK_EINVAL, "step_elems %zu < 512", step);
  if (n == 0) return ops_host && nops ? BRK_OK : brk_fail(BRK_EINVAL, "1..%d ops expected (got %u)", kMaxExprOps, nops);
  static_assert(kMaxExprOps == BRK_EXPR_MAX_OPS && kMaxExprRegs == BRK_EXPR_MAX_REGS && (int)EX_SEL == (int)BRK_EX_SEL, "brk_internal.h / brk.h disagree with brick-device.h");
  int radius = 1;
  if (int rc = brki_expr_check(ops_host, nops, BRK_EXPR_MAX_OPS, result_reg, &radius)) return rc;
  ExprProg pg;
  std::memset(&pg, 0, sizeof pg);
  pg.n = (int)nops, pg.radius = radius, pg.result = (int)result_reg;
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops_host[t];
    ExprOp &d = pg.op[t];
    d.code = (uint8_t)o.code, d.dst = (uint8_t)o.dst, d.a = (uint8_t)o.a, d.b = (uint8_t)o.b, d.c = (uint8_t)o.c;
    const int di = o.offset[0], dj = o.offset[1], dk = o.offset[2];
    d.rel = o.code != EX_TAP ? 0 : (pg.radius == 1 ? tap_rel<1>(di, dj, dk) : (pg.radius == 2 ? tap_rel<2>(di, dj, dk) : (pg.radius == 3 ? tap_rel<3>(di, dj, dk) : tap_rel<4>(di, dj, dk))));
    d.imm = o.imm;
  }
  hipStream_t st = static_cast<hipStream_t>(s);
  const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
  static const char *const names[] = {"k_expr<1,4>", "k_expr<2,4>", "k_expr<3,2>", "k_expr<4,1>"};
  brki_note_kernel(names[pg.radius - 1]);
  if (pg.radius == 1) hipLaunchKernelGGL((k_expr<1, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, pg);
  else if (pg.radius == 2) hipLaunchKernelGGL((k_expr<2, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, pg);
  else if (pg.radius == 3) hipLaunchKernelGGL((k_expr<3, 2>), dim3((((n + 1) / 2 + 7) / 8) * 8), dim3(128), 0, st, adj, in, out, step, ids, first, n, pg);
  else hipLaunchKernelGGL((k_expr<4, 1>), dim3(((n + 7) / 8) * 8), dim3(64), 0, st, adj, in, out, step, ids, first, n, pg);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

extern "C" int brk_stencil_apply_box(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                                     uint32_t base, const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3],
                                     const double *coeff_host, brk_stream s) {
  return brk_stencil_apply_box_ex(id, adj, in, out, step, base, tot, lo, cnt, coeff_host, 0u, s);
}

extern "C" int brk_stencil_apply_box_ex(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                                        uint32_t base, const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3],
                                        const double *coeff_host, uint32_t flags, brk_stream s) {
  if (flags & ~(uint32_t)BRK_SWEEP_REVERSE) return brk_fail(BRK_EINVAL, "brk_stencil_apply_box_ex: flags %#x (only BRK_SWEEP_REVERSE applies to a box)", flags);
  const bool reverse = (flags & BRK_SWEEP_REVERSE) != 0;
  if (!tot || !lo || !cnt) return brk_fail(BRK_EINVAL, "null box geometry");
  for (int a = 0; a < 3; ++a)
    if (lo[a] + cnt[a] > tot[a]) return brk_fail(BRK_EINVAL, "box exceeds the grid on axis %d", a);
  if ((uint64_t)base + (uint64_t)tot[0] * tot[1] * tot[2] > 0xFFFFFFFFull) return brk_fail(BRK_EINVAL, "brick ids overflow 32 bits");
  if (!adj || !in || !out || in == out || step < 512) return brk_fail(BRK_EINVAL, "bad device pointers / step");
  if (((uintptr_t)in | (uintptr_t)out) % 16 || step % 2) return brk_fail(BRK_EINVAL, "brick fields must be 16-byte aligned");
  if (id < 0 || id >= BRK_NSTENCIL) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
  if ((uint64_t)cnt[0] * cnt[1] * cnt[2] == 0) return BRK_OK;
  Coeffs cf;
  if (int rc = make_coeffs(id, coeff_host, cf)) return rc;
  cf.reverse = reverse ? 1u : 0u;
  if (id == BRK_7PT_COEFF || id == BRK_MPI7 || id == BRK_COND7) {
    const uint32_t *regmask = nullptr, *faces = nullptr; // registered adjacency: bricks whose face neighbours are the predicted ones skip their rows
    std::lock_guard<std::mutex> adj_lock(g_adj_mutex);   // until the launch is enqueued
    if (int rc = regular_mask_locked(adj, tot[0], tot[0] * tot[1], 0, static_cast<hipStream_t>(s), &regmask)) return rc;
    if (int rc = face_table_locked(adj, &faces)) return rc;
    return launch_star7_box(id, adj, in, out, step, base, tot, lo, cnt, cf, static_cast<hipStream_t>(s), regmask, faces);
  }
  // other stencils: the same traversal order as an id list, built once per box geometry and kept on the device
  std::lock_guard<std::mutex> lock(g_box_mutex);
  BoxList *e = nullptr;
  // Blocks of 8 rows for every stencil.  Measured for the radius-4 star (round 3, interleaved runs on one box): blocks of 4 rows keep
  // the k-halo lines in the 4-MiB L2 until the brick itself is swept (fabric reads 1.05 -> 0.73 GB, FETCH_SIZE 709 174 KB, traffic
  // 3.18 -> 2.53 GB per 512^3 sweep) but the sweep takes 0.500-0.508 ms instead of 0.493-0.495: the extra reads are served by the
  // Infinity Cache and are not what bounds the kernel, the shorter rows-in-flight window is (profiles/r03_d_25pt_rows_per_block.txt).
  const bool kpair = id == BRK_MPI25 || id == BRK_MPI13; // the stars march two k-consecutive bricks per wave over k-paired lists (k_star_km, brk_starwg.hip)
  if (int rc = box_list_locked(base, tot, lo, cnt, 8u, kpair, id == BRK_MPI25 ? 32u : 0u, &e)) return rc; // radius 4: i-chunks of 32 bricks (see box_list_locked)
  const uint32_t strides[2] = {tot[0], tot[0] * tot[1]};
  if (id == BRK_MPI125 && cnt[0] % 8 == 0 && cube125_q_enabled()) {
    // The 125-point kernel at four waves per SIMD (brk_cube125q.hip) handles groups of four fully linked i-neighbours only.  Over a
    // REGISTERED table that is a provable property of a box: rows are multiples of 8 bricks (every group of four list entries is four
    // consecutive bricks of one row, or four padding entries) and every brick of the box is regular (bit set in the cube-regular
    // mask: all 26 neighbours are the predicted ids).  The second condition is reduced once per list and mask and remembered.
    std::lock_guard<std::mutex> adj_lock(g_adj_mutex); // until the launch is enqueued (lifetime rule of the derived tables)
    const uint32_t *cmask = nullptr;
    uint64_t uid = 0;
    if (int rc = regular_mask_locked(adj, strides[0], strides[1], 1, static_cast<hipStream_t>(s), &cmask, &uid)) return rc;
    if (cmask) {
      if (e->q_uid != uid) {
        bool ok = false;
        if (int rc = all_regular(e->dev, e->n, cmask, static_cast<hipStream_t>(s), &ok)) return rc;
        e->q_uid = uid, e->q_ok = ok;
      }
      if (e->q_ok) {
        if (int rc = launch_cube125_q(adj, in, out, step, e->dev, e->n, cf, static_cast<hipStream_t>(s), strides[0], strides[1], true)) return rc;
        HIP_TRY(hipEventRecord(e->last_use, static_cast<hipStream_t>(s)));
        return BRK_OK;
      }
    }
  }
  if (int rc = launch_stencil(id, adj, in, out, step, e->dev, 0, e->n, coeff_host, s, kpair, reverse, strides)) return rc;
  HIP_TRY(hipEventRecord(e->last_use, static_cast<hipStream_t>(s)));
  return BRK_OK;
}

// for brk_expr_apply_box (brk_jit.hip): the traversal order of a box as a device list (8-row blocks per XCD slab, as for the built-in
// stencils); the cache's mutex stays taken until brki_box_list_unlock has recorded the launch that reads the list
static BoxList *g_box_lent = nullptr;
extern "C" int brki_box_list_lock(uint32_t base, const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], const uint32_t **ids, uint32_t *n) {
  g_box_mutex.lock();
  BoxList *e = nullptr;
  const int rc = box_list_locked(base, tot, lo, cnt, 8u, false, 0u, &e);
  if (rc != BRK_OK) {
    g_box_mutex.unlock();
    return rc;
  }
  g_box_lent = e, *ids = e->dev, *n = e->n;
  return BRK_OK;
}
extern "C" int brki_box_list_unlock(brk_stream s) {
  const hipError_t e = g_box_lent ? hipEventRecord(g_box_lent->last_use, static_cast<hipStream_t>(s)) : hipSuccess;
  g_box_lent = nullptr;
  g_box_mutex.unlock();
  return e == hipSuccess ? BRK_OK : brk_fail(BRK_EHIP, "hipEventRecord: %s", hipGetErrorString(e));
}

extern "C" int brk_link_copy(const brk_link *links, uint32_t n, uint64_t max_bytes, brk_stream s) {
  if (n == 0 || max_bytes == 0) return BRK_OK;
  if (!links) return brk_fail(BRK_EINVAL, "null link table");
  if (n > 65535) return brk_fail(BRK_EINVAL, "%u links exceed one launch (65535)", n);
  if (max_bytes % 16) return brk_fail(BRK_EINVAL, "link sizes must be multiples of 16 bytes");
  const dim3 grid((unsigned)((max_bytes + kLinkChunk - 1) / kLinkChunk), n), block(256);
  hipLaunchKernelGGL(k_link_copy, grid, block, 0, static_cast<hipStream_t>(s), links);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

extern "C" int brk_array_to_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr,
                                  const uint32_t *grid, double *bdat, size_t step, brk_stream s) {
  IterGeom g;
  long nb;
  if (int rc = make_geom(dimlist, padding, ghost, g, nb)) return rc;
  if (!arr || !grid || !bdat) return brk_fail(BRK_EINVAL, "null device pointer");
  hipLaunchKernelGGL(k_iter<kToBrick>, dim3((unsigned)((nb + kWavesPerWG - 1) / kWavesPerWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(s), g, const_cast<double *>(arr), grid, bdat, step, 0.0, nullptr);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

extern "C" int brk_brick_to_array(const long dimlist[3], const long padding[3], const long ghost[3], double *arr,
                                  const uint32_t *grid, const double *bdat, size_t step, brk_stream s) {
  IterGeom g;
  long nb;
  if (int rc = make_geom(dimlist, padding, ghost, g, nb)) return rc;
  if (!arr || !grid || !bdat) return brk_fail(BRK_EINVAL, "null device pointer");
  hipLaunchKernelGGL(k_iter<kToArray>, dim3((unsigned)((nb + kWavesPerWG - 1) / kWavesPerWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(s), g, arr, grid, const_cast<double *>(bdat), step, 0.0, nullptr);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

extern "C" int brk_compare_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr,
                                 const uint32_t *grid, const double *bdat, size_t step, double tol, uint64_t *result,
                                 brk_stream s) {
  IterGeom g;
  long nb;
  if (int rc = make_geom(dimlist, padding, ghost, g, nb)) return rc;
  if (!arr || !grid || !bdat || !result) return brk_fail(BRK_EINVAL, "null device pointer");
  HIP_TRY(hipMemsetAsync(result, 0, 2 * sizeof(uint64_t), static_cast<hipStream_t>(s)));
  hipLaunchKernelGGL(k_iter<kCompare>, dim3((unsigned)((nb + kWavesPerWG - 1) / kWavesPerWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(s), g, const_cast<double *>(arr), grid, const_cast<double *>(bdat), step,
                     tol, reinterpret_cast<unsigned long long *>(result));
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}
