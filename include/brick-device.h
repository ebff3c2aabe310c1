// brick-device.h -- header-only device side of the MI355X brick stencils (gfx950, wave64).
//
// Everything a `__global__` kernel needs to compute ONE brick with ONE wavefront: this is what a rewritten
// `brick("7pt.py", "HIP", (8,8,8), (8,8), b)` line calls (tools/brkscatter), and what libbrk.so's own
// generic kernels are built from (bricklib_amd/csrc/brk_kernels.hip).  The reference counterpart is the text the code
// generator splices in place of the macro (codegen/vecscatter, codegen/st/codegen/backend/cuda.py).
// Contract: called by all 64 lanes of a wave with a wave-uniform brick index; blockDim.x a multiple of 64; the
// function owns a static __shared__ tile per wave of the block.
#ifndef BRICK_DEVICE_H
#define BRICK_DEVICE_H
#ifndef __HIPCC_RTC__ // run-time compilation (brk_expr_compile): hiprtc declares the runtime itself and has no standard headers
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

#include "brk.h"

// (the namespace is not called `brk`: <unistd.h> declares the system call brk(), and the reference's weak/ and strong/ drivers include it)
namespace brklib {

// per-launch parameters every kernel receives by value: the stencil's coefficients and the direction of the sweep
struct Coeffs {
  double c[32];
  // 1: every XCD walks its share of the list / box from the END (workgroup g of the share takes the place of workgroup
  // per_xcd-1-g).  Same bricks, same results; a sweep that starts where the previous one stopped finds the part of the field
  // touched last still in the 256 MB Infinity Cache (brk_stencil_apply_*_ex, BRK_SWEEP_REVERSE)
  uint32_t reverse;
};
// index of this workgroup inside its XCD's share of a launch of 8 * per_xcd workgroups
__device__ __forceinline__ uint32_t wg_in_xcd(const Coeffs &cf, uint32_t per_xcd) {
  const uint32_t g = blockIdx.x >> 3;
  return cf.reverse ? per_xcd - 1u - g : g;
}

// adjacency slot of direction (dk,dj,di), include/brick.h:237-245 of the reference
__host__ __device__ constexpr int adj_slot(int dk, int dj, int di) { return (dk + 1) * 9 + (dj + 1) * 3 + (di + 1); }

// Results leave the chip with the non-temporal hint (`global_store ... nt`): a sweep never re-reads what it writes, so
// the output stream should not displace the input bricks whose faces the following rows and planes still need from L2.
// Measured on the 512^3 7-point sweep: 0.402 -> 0.379 ms (profiles/).
template <class P>
__device__ __forceinline__ void store_result(P *p, double v) {
#ifdef BRK_PLAIN_STORE
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}

// Wave-local LDS ordering.  The hardware runs one wave's DS instructions in order; these only stop the
// compiler from moving a lane's reads above other lanes' writes.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ================================================================================================================
// Star stencils (7-pt, 13-pt, 25-pt): centre + R points along each axis.
// LDS per wave: TJ[8][8+2R][8] (j-extended planes, dense: conflict-free 64-B rows) and TI[8][8][PI]
// (i-extended rows).  PI is chosen so that the four 8-double rows one 32-lane half touches fall on different
// banks (ds_read_b64: 32 double-wide banks): 10 for R=1, 12 for R=2 (4 of 32 lanes 2-way), 24 for R=4.
// ================================================================================================================
template <int R>
struct StarTile {
  static constexpr int T = 8 + 2 * R;
  static constexpr int PI = R == 1 ? 10 : (R == 2 ? 12 : 24);
  static constexpr int TJ_ELEMS = 8 * T * 8;
  static constexpr int TI_ELEMS = 8 * 8 * PI;
  static constexpr int ELEMS = TJ_ELEMS + TI_ELEMS;
  __device__ static constexpr int tj(int k, int jj, int i) { return (k * T + jj) * 8 + i; }
  __device__ static constexpr int ti(int k, int j, int ii) { return TJ_ELEMS + (k * 8 + j) * PI + ii; }
};

enum StarKind { kStar7Coeff, kStarMpi7, kStarMpi13, kStarMpi25, kStarCond7, kStarUser2, kStarUser4 }; // kStarUser*: run-time compiled cell updates (brk_expr_compile)

template <int KIND>
struct StarTraits;
template <> struct StarTraits<kStar7Coeff> { static constexpr int R = 1; };
template <> struct StarTraits<kStarMpi7> { static constexpr int R = 1; };
template <> struct StarTraits<kStarMpi13> { static constexpr int R = 2; };
template <> struct StarTraits<kStarMpi25> { static constexpr int R = 4; };
template <> struct StarTraits<kStarCond7> { static constexpr int R = 1; };
template <> struct StarTraits<kStarUser2> { static constexpr int R = 2; };
template <> struct StarTraits<kStarUser4> { static constexpr int R = 4; };

// One wave computes brick `b` (wave-uniform).  `tile` = this wave's private LDS slice of StarTile<R>::ELEMS doubles.
template <int KIND>
__device__ __forceinline__ void star_brick(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                           double *__restrict__ out, size_t step, uint32_t b, const Coeffs &cf, int lane) {
  constexpr int R = StarTraits<KIND>::R;
  using TL = StarTile<R>;
  const int hi = lane >> 3, lo = lane & 7; // (j,i) when computing; (k,i) / (k,j) when staging faces
  // one coalesced 108-B read of the adjacency row, then lane -> SGPR broadcasts
  const uint32_t arow = adj[(size_t)b * 27 + (lane < 27 ? lane : 13)];
  const uint32_t n_km = __builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0));
  const uint32_t n_kp = __builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0));
  const uint32_t n_jm = __builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0));
  const uint32_t n_jp = __builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0));
  const uint32_t n_im = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1));
  const uint32_t n_ip = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1));

  const double *self = in + (size_t)b * step;
  const double *p_km = in + (size_t)n_km * step, *p_kp = in + (size_t)n_kp * step;
  const double *p_jm = in + (size_t)n_jm * step, *p_jp = in + (size_t)n_jp * step;
  const double *p_im = in + (size_t)n_im * step, *p_ip = in + (size_t)n_ip * step;

  // ---- issue every global load of this brick before touching LDS (all in flight together)
  double col[8 + 2 * R]; // column (.., j, i) along k: [0,R) below, [R,R+8) own, [R+8,R+8+R) above
#pragma unroll
  for (int k = 0; k < 8; ++k) col[R + k] = self[k * 64 + lane];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    col[r] = p_km[(8 - R + r) * 64 + lane];
    col[R + 8 + r] = p_kp[r * 64 + lane];
  }
  double fjm[R], fjp[R], fim[R], fip[R]; // faces: lane = (k=hi, i=lo) for j faces, (k=hi, j=lo) for i faces
#pragma unroll
  for (int r = 0; r < R; ++r) {
    fjm[r] = p_jm[hi * 64 + (8 - R + r) * 8 + lo];
    fjp[r] = p_jp[hi * 64 + r * 8 + lo];
    fim[r] = p_im[hi * 64 + lo * 8 + (8 - R + r)];
    fip[r] = p_ip[hi * 64 + lo * 8 + r];
  }

  // ---- stage the in-plane neighbourhood
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    tile[TL::tj(k, hi + R, lo)] = col[R + k];
    tile[TL::ti(k, hi, lo + R)] = col[R + k];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tile[TL::tj(hi, r, lo)] = fjm[r];
    tile[TL::tj(hi, R + 8 + r, lo)] = fjp[r];
    tile[TL::ti(hi, lo, r)] = fim[r];
    tile[TL::ti(hi, lo, R + 8 + r)] = fip[r];
  }
  wave_lds_sync();

  // ---- compute 8 planes; lane = (j=hi, i=lo)
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double jm[R], jp[R], im[R], ip[R];
#pragma unroll
    for (int r = 1; r <= R; ++r) {
      jm[r - 1] = tile[TL::tj(k, hi + R - r, lo)];
      jp[r - 1] = tile[TL::tj(k, hi + R + r, lo)];
      im[r - 1] = tile[TL::ti(k, hi, lo + R - r)];
      ip[r - 1] = tile[TL::ti(k, hi, lo + R + r)];
    }
    const double c0 = col[R + k];
    double res;
    if (KIND == kStar7Coeff) {
      // stencils/3axis.cpp:60-63 term order
      res = cf.c[5] * col[R + k + 1] + cf.c[6] * col[R + k - 1] + cf.c[3] * jp[0] + cf.c[4] * jm[0] +
            cf.c[1] * ip[0] + cf.c[2] * im[0] + cf.c[0] * c0;
    } else if (KIND == kStarCond7) {
      // stencils/3axis.cpp:104-111 (d3cond): taps clipped at zero, absolute value of the sum
      auto pos = [](double x) { return x > 0.0 ? x : 0.0; };
      const double partial = cf.c[5] * pos(col[R + k + 1]) + cf.c[6] * pos(col[R + k - 1]) + cf.c[3] * pos(jp[0]) +
                             cf.c[4] * pos(jm[0]) + cf.c[1] * pos(ip[0]) + cf.c[2] * pos(im[0]) + cf.c[0] * pos(c0);
      res = partial > 0 ? partial : -partial;
    } else if (KIND == kStarMpi7) {
      // stencils/fake.h:346-349
      res = (col[R + k + 1] + col[R + k - 1] + jp[0] + jm[0] + ip[0] + im[0]) * cf.c[1] + c0 * cf.c[0];
    } else {
      // stencils/fake.h:41-53 (25-pt) / :73-79 (13-pt): outermost shell first, centre last
      res = 0.0;
#pragma unroll
      for (int r = R; r >= 1; --r) {
        const double sh = col[R + k + r] + col[R + k - r] + jp[r - 1] + jm[r - 1] + ip[r - 1] + im[r - 1];
        res = (r == R) ? sh * cf.c[r] : res + sh * cf.c[r];
      }
      res = res + c0 * cf.c[0];
    }
    store_result(&dst[k * 64 + lane], res);
  }
}

// ================================================================================================================
// Cube stencils (27-pt, 125-pt): full (8+2R)^3 tile in LDS.  Faces are staged with structured, coalesced
// loads; the 20 edge/corner blocks through a precomputed per-lane descriptor table.  Compute runs in scatter
// form: each tile value is read once per (dj,di) and FMA'd into the up-to-(2R+1) k-outputs it feeds.
// ================================================================================================================
template <int R>
struct CubeTile {
  static constexpr int T = 8 + 2 * R;
  static constexpr int P = R == 1 ? 10 : (R == 2 ? 12 : T); // i pitch in doubles: some lanes of a half-wave collide 2-way (PMC: half of the
                                             // LDS cycles of the 125-pt kernel are conflict cycles); pitch 24 is conflict-free but halves
                                             // occupancy and measured slower (1.37 vs 1.07 ms).  R = 3, 4 (22 / 32 KiB per wave): stencils
                                             // given as data beyond the 125-point cube, the general path only
  static constexpr int ELEMS = T * T * P;
  __host__ __device__ static constexpr int at(int kk, int jj, int ii) { return (kk * T + jj) * P + ii; }
  static constexpr int NEXTRA = T * T * T - 512 - 6 * 64 * R;
  static constexpr int EXTRA_ITERS = (NEXTRA + 63) / 64;
};

// descriptor = tile index | adjacency slot << 16 | element offset inside the source brick << 21 |
//              i-side tag << 30 (0: cell has di = 0, 1: di = -1, 2: di = +1; lets a kernel skip the cells a sibling provides)
template <int R>
struct ExtraTable {
  uint32_t e[CubeTile<R>::EXTRA_ITERS * 64];
};

template <int R>
constexpr ExtraTable<R> make_extra_table() {
  ExtraTable<R> t{};
  constexpr int T = 8 + 2 * R;
  int n = 0;
  for (int kk = 0; kk < T; ++kk)
    for (int jj = 0; jj < T; ++jj)
      for (int ii = 0; ii < T; ++ii) {
        const int dk = kk < R ? -1 : (kk >= R + 8 ? 1 : 0);
        const int dj = jj < R ? -1 : (jj >= R + 8 ? 1 : 0);
        const int di = ii < R ? -1 : (ii >= R + 8 ? 1 : 0);
        if ((dk != 0) + (dj != 0) + (di != 0) < 2) continue;
        const int sk = (kk - R + 8) % 8, sj = (jj - R + 8) % 8, si = (ii - R + 8) % 8;
        t.e[n++] = (uint32_t)CubeTile<R>::at(kk, jj, ii) | ((uint32_t)adj_slot(dk, dj, di) << 16) |
                   ((uint32_t)(sk * 64 + sj * 8 + si) << 21) | ((uint32_t)(di < 0 ? 1 : (di > 0 ? 2 : 0)) << 30);
      }
  for (; n < CubeTile<R>::EXTRA_ITERS * 64; ++n) t.e[n] = 0xFFFFFFFFu;
  return t;
}

static __device__ const ExtraTable<1> g_extra1 = make_extra_table<1>();
static __device__ const ExtraTable<2> g_extra2 = make_extra_table<2>();
static __device__ const ExtraTable<3> g_extra3 = make_extra_table<3>();
static __device__ const ExtraTable<4> g_extra4 = make_extra_table<4>();

template <int R>
__device__ __forceinline__ const uint32_t *extra_table() {
  static_assert(R >= 1 && R <= 4, "halo radius 1..4");
  if (R == 1) return g_extra1.e;
  if (R == 2) return g_extra2.e;
  if (R == 3) return g_extra3.e;
  return g_extra4.e;
}

enum CubeKind { kCube27, kCube125, kDiamond25, kCubeUser1, kCubeUser2 }; // kCubeUser*: run-time compiled cell updates on the 27- / 125-point kernels' tiles (brk_expr_compile)
template <int KIND> struct CubeTraits;
template <> struct CubeTraits<kCube27> { static constexpr int R = 1; };
template <> struct CubeTraits<kCube125> { static constexpr int R = 2; };
template <> struct CubeTraits<kDiamond25> { static constexpr int R = 2; };
template <> struct CubeTraits<kCubeUser1> { static constexpr int R = 1; };
template <> struct CubeTraits<kCubeUser2> { static constexpr int R = 2; };

// Two sweeps of a 7-point star in one pass: S*S has 25 taps, the points within L1 distance 2.  Tap t of the table below has
// weight Coeffs::c[t]; the launcher fills c[t] = sum over pairs (u, v) of star offsets (incl. 0) with u + v = tap of w(u) w(v).
struct Tap3 {
  int dk, dj, di;
};
constexpr Tap3 kDiamondTaps[25] = {
    {0, 0, 0},  {0, 0, 1},  {0, 0, -1}, {0, 1, 0},  {0, -1, 0}, {0, 0, 2},  {0, 0, -2},  {0, 2, 0},  {0, -2, 0},
    {0, 1, 1},  {0, 1, -1}, {0, -1, 1}, {0, -1, -1},                                                     // dk = 0: 13
    {1, 0, 0},  {1, 0, 1},  {1, 0, -1}, {1, 1, 0},  {1, -1, 0},                                           // dk = +1: 5
    {-1, 0, 0}, {-1, 0, 1}, {-1, 0, -1}, {-1, 1, 0}, {-1, -1, 0},                                         // dk = -1: 5
    {2, 0, 0},  {-2, 0, 0}};

// Scatter form over the planes of an R = 2 tile, fully unrolled with one fence per plane (see cube125_accumulate_static):
// plane kk holds input level k_in = kk - 2 and feeds the outputs k = k_in - dk.  Reads per plane: 13 (a level that feeds an
// output of its own k) or 5 (only |dk| = 1 outputs): 114 per lane instead of 300 for the full cube.  The two outermost
// levels (k_in = -2 and 9) only ever meet the centre tap of |dk| = 2, i.e. the lane's own column: they stay in registers
// (below, above) and the tile holds the 10 planes kk = 1..10 (rowp points at the virtual plane 0).
// in-plane positions (dj, di) a plane can be asked for: the first 5 serve the |dk| = 1 taps, all 13 the dk = 0 taps
struct Pos2 {
  int dj, di;
};
constexpr Pos2 kDiamondPos[13] = {{0, 0}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}, {0, 2}, {0, -2}, {2, 0}, {-2, 0}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
constexpr int diamond_pos_of_tap(int t) { return t < 13 ? t : (t < 18 ? t - 13 : (t < 23 ? t - 18 : 0)); }
constexpr int diamond_reads_of_plane(int kk) { return (kk >= 2 && kk <= 9) ? 13 : 5; } // levels 0..7 feed their own output

template <int PS, int RS, bool PIPELINED, class DST>
__device__ __forceinline__ void diamond25_accumulate(const double *rowp, const int (&co)[5], const Coeffs &cf, DST dst, int lane,
                                                     double below, double above) {
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  acc[0] = cf.c[24] * below; // tap (-2,0,0): out[0] reads level -2
  const double *colp[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) colp[d] = rowp + co[d];
  double v[2][13];
  auto load = [&](int kk, double (&dstv)[13]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 13; ++q)
      if (q < diamond_reads_of_plane(kk)) dstv[q] = colp[kDiamondPos[q].di + 2][kk * PS + kDiamondPos[q].dj * RS];
  };
  if (PIPELINED) load(1, v[1]);
#pragma unroll
  for (int kk = 1; kk < 11; ++kk) {
    if (PIPELINED) {
      if (kk + 1 < 11) load(kk + 1, v[(kk + 1) & 1]); // the next plane's reads are in flight during this plane's arithmetic
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load(kk, v[kk & 1]);
    }
    const int kin = kk - 2;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
      const int k = kin - kDiamondTaps[t].dk; // output this tap of this plane feeds
      if (k >= 0 && k < 8) acc[k] += cf.c[t] * v[kk & 1][diamond_pos_of_tap(t)];
    }
    if (kk >= 4) store_result(&dst[(kk - 4) * 64 + lane], acc[kk - 4]); // output kk-4 has received its last plane
    __builtin_amdgcn_sched_barrier(0);
  }
  store_result(&dst[7 * 64 + lane], acc[7] + cf.c[23] * above); // tap (+2,0,0): out[7] reads level 9
}

// coefficient class of a 125-pt offset: sorted(|di|,|dj|,|dk|) -> 0..9 (stencils/mpi125pt.py:12-33)
__host__ __device__ constexpr int cls125(int a, int b, int c) {
  a = a < 0 ? -a : a;
  b = b < 0 ? -b : b;
  c = c < 0 ? -c : c;
  const int lo = a < b ? (a < c ? a : c) : (b < c ? b : c);
  const int hi = a > b ? (a > c ? a : c) : (b > c ? b : c);
  const int mid = a + b + c - lo - hi;
  // (lo,mid,hi): 000 001 002 011 012 022 111 112 122 222
  return lo == 0 ? (mid == 0 ? hi : (mid == 1 ? 2 + hi : 5)) : (lo == 1 ? (mid == 1 ? 5 + hi : 8) : 9);
}

// 125-point accumulation with mirror-image taps folded (the weights depend only on the sorted |offsets|, 10 classes,
// stencils/mpi125pt.py:12-33): per tile plane, add the +-di pairs (2 adds per row, 5 rows), then the +-dj pairs (6 adds);
// the 9 values u[a][b] (a = |di|, b = |dj|) feed the <=5 outputs k = kk-2-dk with weight C[class(|dk|,b,a)].
// ~730 fp64 VALU ops per lane per brick instead of 1000 FMAs; LDS reads unchanged (300).
// The plane loop is a real loop over a 5-entry ring of partial outputs: fully unrolled, hipcc hoists the LDS reads of all 12
// planes (256 VGPRs + 256 AGPRs + scratch).  rowp: per-lane pointer to tile(plane 0, row j+R, column 0);
// co[d]: per-lane column offset of tap di = d-2 (own tile, or a sibling's tile 8 columns away); PS/RS: plane/row pitch.
template <int PS, int RS>
__device__ __forceinline__ void cube125_load_plane(double (&v)[25], const double *rowp, const int (&co)[5]) {
#pragma unroll
  for (int dj = 0; dj < 5; ++dj)
#pragma unroll
    for (int d = 0; d < 5; ++d) v[dj * 5 + d] = rowp[(dj - 2) * RS + co[d]];
}

// fold one plane (25 values, [dj][di]) and add its contributions to the ring of partial outputs
__device__ __forceinline__ void cube125_plane(const double (&v)[25], double (&ring)[5], int kk, const Coeffs &cf) {
  double u[3][3];
#pragma unroll
  for (int dj = 0; dj < 5; ++dj) {
    const double s0 = v[dj * 5 + 2], s1 = v[dj * 5 + 1] + v[dj * 5 + 3], s2 = v[dj * 5] + v[dj * 5 + 4];
    const int bcls = dj < 2 ? 2 - dj : dj - 2;
    if (dj <= 2) { // first visit of this |dj| class
      u[0][bcls] = s0;
      u[1][bcls] = s1;
      u[2][bcls] = s2;
    } else {
      u[0][bcls] += s0;
      u[1][bcls] += s1;
      u[2][bcls] += s2;
    }
  }
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    if ((unsigned)(kk - 4 + m) < 8u) { // wave-uniform: the output exists
      const int c = m < 2 ? 2 - m : m - 2; // |dk|
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) ring[m] += cf.c[cls125(c, bb, a)] * u[a][bb];
    }
  }
}

__device__ __forceinline__ void cube125_retire(double (&ring)[5], int kk, double *dst, int lane) {
  if (kk >= 4) store_result(&dst[(kk - 4) * 64 + lane], ring[0]); // output kk-4 has received its last plane
#pragma unroll
  for (int m = 0; m < 4; ++m) ring[m] = ring[m + 1];
  ring[4] = 0.0;
}

// Plane loop, software-pipelined by hand: the 25 LDS reads of plane kk+1 are issued before plane kk is folded, with two
// named value sets (no copies); 6 trips x 2 planes.
template <int PS, int RS>
__device__ __forceinline__ void cube125_accumulate(const double *rowp, const int (&co)[5], const Coeffs &cf,
                                                   double *dst, int lane) {
  double ring[5] = {0.0, 0.0, 0.0, 0.0, 0.0}; // ring[m] = partial output k = kk-4+m
  double va[25], vb[25];
  cube125_load_plane<PS, RS>(va, rowp, co);
#pragma unroll 1
  for (int kk = 0; kk < 12; kk += 2) {
    cube125_load_plane<PS, RS>(vb, rowp + PS, co);
    __builtin_amdgcn_sched_barrier(0); // keep the reads of the next plane ahead of this plane's arithmetic
    cube125_plane(va, ring, kk, cf);
    cube125_retire(ring, kk, dst, lane);
    if (kk + 2 < 12) cube125_load_plane<PS, RS>(va, rowp + 2 * PS, co);
    __builtin_amdgcn_sched_barrier(0);
    cube125_plane(vb, ring, kk + 1, cf);
    cube125_retire(ring, kk + 1, dst, lane);
    rowp += 2 * PS;
  }
}

// Same without the hand pipelining (one value set, ~100 VGPRs less): for kernels that hide the LDS latency with
// more resident waves instead.
template <int PS, int RS>
__device__ __forceinline__ void cube125_accumulate_simple(const double *rowp, const int (&co)[5], const Coeffs &cf,
                                                          double *dst, int lane) {
  double ring[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
  for (int kk = 0; kk < 12; ++kk) {
    double v[25];
    cube125_load_plane<PS, RS>(v, rowp, co);
    cube125_plane(v, ring, kk, cf);
    cube125_retire(ring, kk, dst, lane);
    rowp += PS;
  }
}

// Fully unrolled plane loop with one scheduling fence per plane: the ring becomes eight statically named accumulators
// (no register rotation: 17 moves per plane less), the 25 reads of a plane take immediate offsets from five per-lane column
// bases (no per-plane address adds), and the validity tests of the outputs are compile-time.  The fences keep hipcc from
// hoisting the reads of all planes (which costs 256 VGPRs + scratch, see above); one plane's 25 values are live at a time.
template <int PS, int RS>
__device__ __forceinline__ void cube125_accumulate_static(const double *rowp, const int (&co)[5], const Coeffs &cf,
                                                          double *dst, int lane) {
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  const double *colp[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) colp[d] = rowp + co[d];
#pragma unroll
  for (int kk = 0; kk < 12; ++kk) {
    double u[3][3];
#pragma unroll
    for (int dj = 0; dj < 5; ++dj) {
      const double v0 = colp[0][kk * PS + (dj - 2) * RS], v1 = colp[1][kk * PS + (dj - 2) * RS];
      const double v2 = colp[2][kk * PS + (dj - 2) * RS], v3 = colp[3][kk * PS + (dj - 2) * RS];
      const double v4 = colp[4][kk * PS + (dj - 2) * RS];
      const double s0 = v2, s1 = v1 + v3, s2 = v0 + v4;
      const int bcls = dj < 2 ? 2 - dj : dj - 2;
      if (dj <= 2) {
        u[0][bcls] = s0;
        u[1][bcls] = s1;
        u[2][bcls] = s2;
      } else {
        u[0][bcls] += s0;
        u[1][bcls] += s1;
        u[2][bcls] += s2;
      }
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const int k = kk - 4 + m; // output fed through |dk| = |2 - m|
      if (k >= 0 && k < 8) {
        const int c = m < 2 ? 2 - m : m - 2;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) acc[k] += cf.c[cls125(c, bb, a)] * u[a][bb];
      }
    }
    if (kk >= 4) store_result(&dst[(kk - 4) * 64 + lane], acc[kk - 4]); // output kk-4 has received its last plane
    __builtin_amdgcn_sched_barrier(0); // a fence every second plane instead: 140 VGPRs and 1.08 instead of 0.70 ms
  }
}

// Software-pipelined form of the same accumulation.  What the counters said about cube125_accumulate_static (profiles/
// r01_j_pmc_125pt_512.txt): the LDS pipe is busy 56 % and the fp64 pipe 53 % of the kernel, one after the other -- each plane
// is 25 ds_read_b64 in a burst, a full wait, then ~61 VALU operations, and three waves per SIMD do not interleave that by
// themselves.  Here ONE wave overlaps the two pipes: the plane loop is cut into 60 row steps (plane kk, row dj); step r
//   (1) issues the 5 reads of row r + D                     (D rows = 5 D values in flight, counted lgkmcnt waits),
//   (2) multiplies one output's share of the PREVIOUS plane (its folded values are complete): 6 FMAs,
//   (3) folds row r into the current plane's values         (2 + 3 adds),
// so a read has ~D x 56 cycles of fp64 work between issue and use.  Further, the class weights are symmetric in (|di|,|dj|)
// too (stencils/mpi125pt.py:12-33: the class is the SORTED triple), so the folded plane u[a][b] is folded once more,
// u[a][b] + u[b][a], and an output takes 6 FMAs per plane instead of 9: 19 adds per plane + 6 FMAs per (plane, output)
// = 468 fp64 operations per lane instead of 552.  Summation order differs from the other forms in the last bits only.
template <int PS, int RS, int D = 3>
__device__ __forceinline__ void cube125_accumulate_rows(const double *rowp, const int (&co)[5], const Coeffs &cf,
                                                        double *dst, int lane) {
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  const double *colp[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) colp[d] = rowp + co[d];
  double v[D + 1][5]; // rows in flight
  double un[3][3];    // plane being folded: un[a][b], a = |di|, b = |dj|
  double w[6];        // previous plane, folded in (a, b) too: (0,0) (1,1) (2,2) (0,1) (0,2) (1,2)
  constexpr int wa[6] = {0, 1, 2, 0, 0, 1}, wb[6] = {0, 1, 2, 1, 2, 2};
  auto load_row = [&](int r) __attribute__((always_inline)) {
    const int kk = r / 5, dj = r % 5 - 2;
#pragma unroll
    for (int d = 0; d < 5; ++d) v[r % (D + 1)][d] = colp[d][kk * PS + dj * RS];
  };
  // output k = pl - 4 + m receives plane pl through |dk| = |2 - m|
  auto fma_chunk = [&](int pl, int m) __attribute__((always_inline)) {
    const int k = pl - 4 + m;
    if (k >= 0 && k < 8) {
      const int c = m < 2 ? 2 - m : m - 2;
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[k] += cf.c[cls125(c, wb[q], wa[q])] * w[q];
    }
    if (m == 0 && pl >= 4) store_result(&dst[(pl - 4) * 64 + lane], acc[pl - 4]); // output pl-4 has received its last plane
  };
#pragma unroll
  for (int r = 0; r < D; ++r) load_row(r);
#pragma unroll
  for (int r = 0; r < 60; ++r) {
    const int pl = r / 5, q = r % 5;
    if (r + D < 60) load_row(r + D);
    __builtin_amdgcn_sched_barrier(0);
    if (pl > 0) fma_chunk(pl - 1, 4 - q); // m = 4 (the newest output) first, m = 0 (retired and stored) last
    {
      const double(&x)[5] = v[r % (D + 1)];
      const double s0 = x[2], s1 = x[1] + x[3], s2 = x[0] + x[4];
      const int bcls = q < 2 ? 2 - q : q - 2;
      if (q <= 2) {
        un[0][bcls] = s0, un[1][bcls] = s1, un[2][bcls] = s2;
      } else {
        un[0][bcls] += s0, un[1][bcls] += s1, un[2][bcls] += s2;
      }
    }
    if (q == 4) { // plane complete: fold (a, b) <-> (b, a) and hand it to the multiplier
      w[0] = un[0][0], w[1] = un[1][1], w[2] = un[2][2];
      w[3] = un[0][1] + un[1][0], w[4] = un[0][2] + un[2][0], w[5] = un[1][2] + un[2][1];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int m = 4; m >= 0; --m) fma_chunk(11, m);
}

// Output-side form of the 125-point accumulation.  The class weights depend on sorted(|di|,|dj|,|dk|) only
// (stencils/mpi125pt.py:12-33), so
//   out(k,j,i) = P0(k,j,i) + P1(k,j-1,i) + P1(k,j+1,i) + P2(k,j-2,i) + P2(k,j+2,i),
//   Pe(k,j',i) = sum over (|di|,|dk|) of c(|di|,e,|dk|) x(k+dk, j', i+di)
// -- the j-shift is applied ONCE per output to the finished partial sums instead of to every input row.  A lane (j', i)
// then needs only row j' of each plane: 5 LDS reads per plane, 60 per lane instead of 300, and (with the (|di|,|dk|)
// fold) 27 fp64 operations per output.  The partial sums are also needed in the four halo rows j' = -2,-1,8,9: 8 k x 4 rows
// x 8 i = 256 positions = one more pass in which the lower half of the wave takes k = 0..3 and the upper half k = 4..7
// (40 reads, 100 operations).  The shift itself goes through a 12-row x 8-column slot per e in LDS that only this wave
// touches (LDS operations of one wave execute in order, so write -> shifted read needs no wait in between): 4 writes and
// 4 reads per output.  Per lane: ~165 LDS operations and ~375 fp64 operations instead of 300 and 468.
// `tile` = the wave's dense tile D[12][12][8] (plane pitch PS, row pitch RS); co[d] = offset of column i+d-2 from the row
// start (own tile or the slot on that side); `slot` = 192 doubles of LDS private to the wave.
template <int PS, int RS, int NP, bool MAIN, class Emit>
__device__ __forceinline__ void cube125_jout_stream(const double *rowp, const int (&co)[5], const Coeffs &cf, Emit emit) {
  constexpr int wa[6] = {0, 1, 2, 0, 0, 1}, wb[6] = {0, 1, 2, 1, 2, 2};
  double x[2][5]; // rows in flight
  double V[5][3]; // |di|-folded rows of the last five planes
  auto load = [&](int p) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < 5; ++d) x[p & 1][d] = rowp[p * PS + co[d]];
  };
  load(0);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p + 1 < NP) load(p + 1);
    __builtin_amdgcn_sched_barrier(0);
    V[p % 5][0] = x[p & 1][2], V[p % 5][1] = x[p & 1][1] + x[p & 1][3], V[p % 5][2] = x[p & 1][0] + x[p & 1][4];
    if (p >= 4) { // output p-4: planes p-4 .. p, centre p-2
      double W[3][3], G[6];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        W[0][d] = V[(p - 2) % 5][d];
        W[1][d] = V[(p - 3) % 5][d] + V[(p - 1) % 5][d];
        W[2][d] = V[(p - 4) % 5][d] + V[p % 5][d];
      }
      G[0] = W[0][0], G[1] = W[1][1], G[2] = W[2][2];
      G[3] = W[0][1] + W[1][0], G[4] = W[0][2] + W[2][0], G[5] = W[1][2] + W[2][1];
      double b[3];
#pragma unroll
      for (int e = MAIN ? 0 : 1; e < 3; ++e) {
        b[e] = cf.c[cls125(e, wa[0], wb[0])] * G[0];
#pragma unroll
        for (int q = 1; q < 6; ++q) b[e] += cf.c[cls125(e, wa[q], wb[q])] * G[q];
      }
      emit(p - 4, MAIN ? b[0] : 0.0, b[1], b[2]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ALIASED: the shift slots are not memory of their own but columns 2..5 of planes 0..3 of the wave's OWN tile (`slot` = tile): those
// columns are read by no other wave (neighbours read columns 0, 1 / 6, 7 only) and this wave has finished with planes 0..3 -- halo
// pass first, main-pass planes in order -- before output 0 is emitted behind plane 4.  Slot (row r, column c) -> plane c / 4 (s1) or
// 2 + c / 4 (s2), row r, column 2 + c % 4; every plane and tile starts at a multiple of 32 doubles, so lanes c and c + 4 of a row
// meet in one bank (2-way conflict on the 4 + 4 shift accesses per output): the price of 6 KiB of LDS per group.
template <int PS, int RS, bool ALIASED = false>
__device__ __forceinline__ void cube125_accumulate_jout(const double *tile, const int (&co)[5], const Coeffs &cf, double *slot,
                                                        double *dst, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  const int kh = lane >> 5, jh = (lane >> 3) & 3;
  const int hrow = jh < 2 ? jh : jh + 8; // halo rows 0, 1, 10, 11 of the tile
  double h1[4], h2[4];
  cube125_jout_stream<PS, RS, 8, false>(tile + kh * 4 * PS + hrow * RS, co, cf,
                                        [&](int k, double, double b1, double b2) __attribute__((always_inline)) {
                                          asm volatile("" : "+v"(b1), "+v"(b2)); // computed here, by all lanes, not sunk into the half-wave branch that uses it
                                          h1[k] = b1, h2[k] = b2;
                                        });
  double *s1 = ALIASED ? slot + (lo >> 2) * PS + 2 + (lo & 3) : slot + lo, *s2 = ALIASED ? s1 + 2 * PS : slot + 96 + lo;
  double pb0 = 0.0, r[4] = {0.0, 0.0, 0.0, 0.0};
  auto retire = [&](int k) __attribute__((always_inline)) { store_result(&dst[k * 64 + lane], pb0 + ((r[0] + r[1]) + (r[2] + r[3]))); };
  cube125_jout_stream<PS, RS, 12, true>(tile + (hi + 2) * RS, co, cf, [&](int k, double b0, double b1, double b2) __attribute__((always_inline)) {
    if (k > 0) retire(k - 1);
    wave_lds_sync(); // the shifted reads of output k-1 stay in front of these writes
    s1[(hi + 2) * 8] = b1, s2[(hi + 2) * 8] = b2;
    if (kh == (k >> 2)) s1[hrow * 8] = h1[k & 3], s2[hrow * 8] = h2[k & 3];
    wave_lds_sync();
    r[0] = s1[(hi + 1) * 8], r[1] = s1[(hi + 3) * 8], r[2] = s2[hi * 8], r[3] = s2[(hi + 4) * 8];
    pb0 = b0;
  });
  retire(7);
}

// Load brick `b` (wave-uniform) with its full halo of radius R into `tile` (this wave's private LDS slice of
// CubeTile<R>::ELEMS doubles): own brick and the 6 faces by structured coalesced loads, edge/corner cells through the
// descriptor table.
template <int R>
__device__ __forceinline__ void cube_stage(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                           size_t step, uint32_t b, int lane) {
  using TL = CubeTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  // The gather descriptors depend on the lane only and are requested first: vector loads return in order, so a table load
  // issued behind the brick's own loads would come back after them, and the gathers -- which need it for their addresses --
  // could not even be issued until a full round trip of the brick had passed (in-kernel stamps).
  const uint32_t *xt = extra_table<R>();
  uint32_t xd[TL::EXTRA_ITERS];
#pragma unroll
  for (int t = 0; t < TL::EXTRA_ITERS; ++t) xd[t] = xt[t * 64 + lane];
  const uint32_t arow = adj[(size_t)b * 27 + (lane < 27 ? lane : 13)];
  const double *self = in + (size_t)b * step;
  const double *p_km = in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)) * step;
  const double *p_kp = in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0)) * step;
  const double *p_jm = in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0)) * step;
  const double *p_jp = in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0)) * step;
  const double *p_im = in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1)) * step;
  const double *p_ip = in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1)) * step;

  // ---- loads: own brick, 6 faces, then edge/corner cells via descriptors
  double own[8], fkm[R], fkp[R], fjm[R], fjp[R], fim[R], fip[R];
#pragma unroll
  for (int k = 0; k < 8; ++k) own[k] = self[k * 64 + lane];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    fkm[r] = p_km[(8 - R + r) * 64 + lane];
    fkp[r] = p_kp[r * 64 + lane];
    fjm[r] = p_jm[hi * 64 + (8 - R + r) * 8 + lo];
    fjp[r] = p_jp[hi * 64 + r * 8 + lo];
    fim[r] = p_im[hi * 64 + lo * 8 + (8 - R + r)];
    fip[r] = p_ip[hi * 64 + lo * 8 + r];
  }
  double xv[TL::EXTRA_ITERS];
#pragma unroll
  for (int t = 0; t < TL::EXTRA_ITERS; ++t) {
    const bool live = xd[t] != 0xFFFFFFFFu;
    // neighbour id of this lane's cell: pull lane (slot) of the adjacency-row register
    const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((xd[t] >> 16) & 31u) << 2), (int)arow);
    xv[t] = live ? in[(size_t)nb * step + ((xd[t] >> 21) & 511u)] : 0.0;
  }

  // ---- stage
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[TL::at(k + R, hi + R, lo + R)] = own[k];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tile[TL::at(r, hi + R, lo + R)] = fkm[r];
    tile[TL::at(R + 8 + r, hi + R, lo + R)] = fkp[r];
    tile[TL::at(hi + R, r, lo + R)] = fjm[r];
    tile[TL::at(hi + R, R + 8 + r, lo + R)] = fjp[r];
    tile[TL::at(hi + R, lo + R, r)] = fim[r];
    tile[TL::at(hi + R, lo + R, R + 8 + r)] = fip[r];
  }
#pragma unroll
  for (int t = 0; t < TL::EXTRA_ITERS; ++t)
    if (xd[t] != 0xFFFFFFFFu) tile[xd[t] & 0xFFFFu] = xv[t];
  wave_lds_sync();
}

// One wave computes brick `b` (wave-uniform).  `tile` = this wave's private LDS slice of CubeTile<R>::ELEMS doubles.
template <int KIND>
__device__ __forceinline__ void cube_brick(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                           double *__restrict__ out, size_t step, uint32_t b, const Coeffs &cf, int lane) {
  constexpr int R = CubeTraits<KIND>::R;
  using TL = CubeTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  cube_stage<R>(tile, adj, in, step, b, lane);

  // ---- scatter-form accumulate; lane = (j=hi, i=lo)
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  if (KIND == kCube125) {
    const int co[5] = {lo + R - 2, lo + R - 1, lo + R, lo + R + 1, lo + R + 2};
    cube125_accumulate<TL::T * TL::P, TL::P>(tile + (hi + R) * TL::P, co, cf, out + (size_t)b * step, lane);
    return;
  } else {
#pragma unroll
    for (int kk = 0; kk < TL::T; ++kk) {
#pragma unroll
      for (int dj = -R; dj <= R; ++dj)
#pragma unroll
        for (int di = -R; di <= R; ++di) {
          const double v = tile[TL::at(kk, hi + R + dj, lo + R + di)];
#pragma unroll
          for (int dk = -R; dk <= R; ++dk) {
            const int k = kk - R - dk; // output plane fed through tap dk
            if (k >= 0 && k < 8) acc[k] += cf.c[adj_slot(dk, dj, di)] * v;
          }
        }
    }
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}


// ---------------------------------------------------------------------------------------------------------------
// Linear stencils given as data: out[p] = sum_t w[t] * in[p + (di,dj,dk)[t]], any subset of the 125-point cube
// (|offset| <= 2 per axis), taps applied in list order.  The table travels as a kernel argument (1.5 KB).
// rel[t] = offset of tap t relative to the centre cell inside a CubeTile<R>: (dk * T + dj) * P + di.
// One LDS read + one FMA per tap and output plane: the general path, slower than the hand-scheduled kernels above.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxTaps = 125;     // table as a kernel argument (radius <= 2)
constexpr int kMaxFarTaps = 729;  // table in device memory (radius 3, 4)
struct Taps {
  int n, radius;
  int rel[kMaxTaps];
  double w[kMaxTaps];
};

template <int R>
__host__ __device__ constexpr int tap_rel(int di, int dj, int dk) { return (dk * CubeTile<R>::T + dj) * CubeTile<R>::P + di; }

template <int R>
__device__ __forceinline__ void taps_brick(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                           double *__restrict__ out, size_t step, uint32_t b, const Taps &tp, int lane) {
  using TL = CubeTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  cube_stage<R>(tile, adj, in, step, b, lane);
  const double *c = tile + TL::at(R, hi + R, lo + R);
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int t = 0; t < tp.n; ++t) {
    const int o = tp.rel[t];
    const double w = tp.w[t];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += w * c[k * TL::T * TL::P + o];
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}

// Cube-shaped tables of radius 3 and 4 (up to 9^3 = 729 taps): the table lives in device memory, read through scalar loads.
struct TapsFar {
  int n;
  const int *rel;    // tap_rel<R>
  const double *w;
};

template <int R>
__device__ __forceinline__ void taps_far_brick(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                               double *__restrict__ out, size_t step, uint32_t b, const TapsFar &tp, int lane) {
  using TL = CubeTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  cube_stage<R>(tile, adj, in, step, b, lane);
  const double *c = tile + TL::at(R, hi + R, lo + R);
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int t = 0; t < tp.n; ++t) {
    const int o = __builtin_amdgcn_readfirstlane(tp.rel[t]);
    const double w = tp.w[t];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += w * c[k * TL::T * TL::P + o];
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}

// waves of one workgroup that fit the 64 KiB of static LDS with a CubeTile<R> each
template <int R>
struct CubeWaves {
  static constexpr int WAVES = R <= 2 ? 4 : (R == 3 ? 2 : 1); // (not `N`: the reference's drivers #define N)
};

// user-kernel entry point (what tools/brkscatter emits for a stencil script that is not one of the built-in ones)
template <int R, int MAXW = CubeWaves<R>::WAVES>
__device__ __forceinline__ void brick_apply_taps(const uint32_t *adj, const double *in, double *out, size_t step, uint32_t b,
                                                 const Taps &tp) {
  __shared__ double lds[MAXW * CubeTile<R>::ELEMS];
  taps_brick<R>(lds + (threadIdx.x >> 6) * CubeTile<R>::ELEMS, adj, in, out, step, __builtin_amdgcn_readfirstlane(b), tp,
                threadIdx.x & 63);
}

// The same with the tap table known at compile time (NTAPS taps; drivers of the reference #define N, offsets `rel`, weights `w` in registers): fully unrolled.
template <int R, int NTAPS, int MAXW = CubeWaves<R>::WAVES>
__device__ __forceinline__ void brick_apply_taps_static(const uint32_t *adj, const double *in, double *out, size_t step,
                                                        uint32_t b, const int (&rel)[NTAPS], const double (&w)[NTAPS]) {
  using TL = CubeTile<R>;
  __shared__ double lds[MAXW * TL::ELEMS];
  const int lane = threadIdx.x & 63, hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (threadIdx.x >> 6) * TL::ELEMS;
  b = __builtin_amdgcn_readfirstlane(b);
  cube_stage<R>(tile, adj, in, step, b, lane);
  const double *c = tile + TL::at(R, hi + R, lo + R);
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += w[t] * c[k * TL::T * TL::P + rel[t]];
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}

// Hundreds of taps known at compile time (a cube-shaped script of radius 3 or 4: up to 9^3): a weight table in registers would
// spill, so lane 0 of each wave writes offsets and weights to LDS behind the tile -- every weight is evaluated and stored at once --
// and the tap loop stays rolled: one broadcast LDS read of (offset, weight) per tap, then the 8 plane reads + FMAs.
template <int R, int NTAPS, int MAXW = CubeWaves<R>::WAVES>
__device__ __forceinline__ void brick_apply_taps_table(const uint32_t *adj, const double *in, double *out, size_t step,
                                                       uint32_t b, const int (&rel)[NTAPS], const double (&w)[NTAPS]) {
  using TL = CubeTile<R>;
  static_assert(MAXW * (TL::ELEMS + NTAPS + (NTAPS + 1) / 2) * 8 <= 65536, "tile + tap table exceed the static LDS of a workgroup");
  __shared__ double lds[MAXW * TL::ELEMS];
  __shared__ double wtab[MAXW * NTAPS];
  __shared__ int rtab[MAXW * NTAPS];
  const int lane = threadIdx.x & 63, hi = lane >> 3, lo = lane & 7, wave = threadIdx.x >> 6;
  double *tile = lds + wave * TL::ELEMS;
  double *wt = wtab + wave * NTAPS;
  int *rt = rtab + wave * NTAPS;
  b = __builtin_amdgcn_readfirstlane(b);
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) wt[t] = w[t], rt[t] = rel[t];
  }
  cube_stage<R>(tile, adj, in, step, b, lane); // ends with the wave's LDS fence: the table is visible too
  const double *c = tile + TL::at(R, hi + R, lo + R);
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int t = 0; t < NTAPS; ++t) {
    const int o = __builtin_amdgcn_readfirstlane(rt[t]);
    const double wv = wt[t];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += wv * c[k * TL::T * TL::P + o];
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}

// ---------------------------------------------------------------------------------------------------------------
// NON-LINEAR stencils given as data (round 5): the reference's generator takes any expression over grid references, constants,
// functions and If (codegen/st/codegen/base.py:678; stencils/cond.py uses max and If).  Here such a script arrives as a small
// register program (bricklib_amd/spec.py reduces the script to it; tools/brkscatter writes it next to a rewritten brick(...) line):
// every op names a destination register, up to three source registers, an immediate or a tap.  One wave stages the brick with its
// radius-R halo (cube_stage) and every lane runs the program once per k-plane; the op loop is wave-uniform, the register file is a
// per-lane array indexed by uniform values.  The general path: one LDS read per tap, a handful of instructions per op.
// ---------------------------------------------------------------------------------------------------------------
enum { EX_TAP = 0, EX_CONST, EX_ADD, EX_SUB, EX_MUL, EX_DIV, EX_NEG, EX_MAX, EX_MIN, EX_ABS, EX_SQRT, EX_GT, EX_GE, EX_LT, EX_LE, EX_EQ, EX_NE, EX_SEL, EX_NOPS };
constexpr int kMaxExprOps = 120, kMaxExprRegs = 16;
struct ExprOp {
  uint8_t code, dst, a, b, c, pad_[3];
  int32_t rel; // EX_TAP: offset of the tap relative to the centre cell inside a CubeTile<R> (tap_rel<R>)
  double imm;  // EX_CONST
};
struct ExprProg {
  int32_t n, radius, result, pad_;
  ExprOp op[kMaxExprOps];
};
static_assert(sizeof(ExprProg) <= 3072, "the program travels as a kernel argument");

template <int R>
__device__ __forceinline__ void expr_brick(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out,
                                           size_t step, uint32_t b, const ExprProg &pg, int lane) {
  using TL = CubeTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  cube_stage<R>(tile, adj, in, step, b, lane);
  const double *c = tile + TL::at(R, hi + R, lo + R);
  double *dst = out + (size_t)b * step;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    double reg[kMaxExprRegs];
#pragma unroll 1
    for (int t = 0; t < pg.n; ++t) {
      const ExprOp o = pg.op[t];
      const double x = reg[o.a & (kMaxExprRegs - 1)], y = reg[o.b & (kMaxExprRegs - 1)], z = reg[o.c & (kMaxExprRegs - 1)];
      double v;
      switch (o.code) { // wave-uniform
      case EX_TAP: v = c[k * TL::T * TL::P + o.rel]; break;
      case EX_CONST: v = o.imm; break;
      case EX_ADD: v = x + y; break;
      case EX_SUB: v = x - y; break;
      case EX_MUL: v = x * y; break;
      case EX_DIV: v = x / y; break;
      case EX_NEG: v = -x; break;
      case EX_MAX: v = x > y ? x : y; break; // the C macro / fmax of the reference's generated code on ordered values
      case EX_MIN: v = x < y ? x : y; break;
      case EX_ABS: v = x < 0 ? -x : x; break;
      case EX_SQRT: v = __builtin_sqrt(x); break;
      case EX_GT: v = x > y ? 1.0 : 0.0; break;
      case EX_GE: v = x >= y ? 1.0 : 0.0; break;
      case EX_LT: v = x < y ? 1.0 : 0.0; break;
      case EX_LE: v = x <= y ? 1.0 : 0.0; break;
      case EX_EQ: v = x == y ? 1.0 : 0.0; break;
      case EX_NE: v = x != y ? 1.0 : 0.0; break;
      default: v = x != 0.0 ? y : z; break; // EX_SEL: If(cond, then, else)
      }
      reg[o.dst & (kMaxExprRegs - 1)] = v;
    }
    store_result(&dst[k * 64 + lane], reg[pg.result & (kMaxExprRegs - 1)]);
  }
}

// user-kernel entry point (what tools/brkscatter emits for a non-linear stencil script)
template <int R, int MAXW = CubeWaves<R>::WAVES>
__device__ __forceinline__ void brick_apply_expr(const uint32_t *adj, const double *in, double *out, size_t step, uint32_t b, const ExprProg &pg) {
  __shared__ double lds[MAXW * CubeTile<R>::ELEMS];
  expr_brick<R>(lds + (threadIdx.x >> 6) * CubeTile<R>::ELEMS, adj, in, out, step, __builtin_amdgcn_readfirstlane(b), pg, threadIdx.x & 63);
}

// The same with the expression known at compile time (what tools/brkscatter emits for a non-linear stencil script: the script's operations
// written out as straight-line C++ in a lambda).  f(c): the value of one cell, c = the cell's own position in the staged tile, a
// grid reference in[p + (di, dj, dk)] is c[tap_rel<R>(di, dj, dk)].
template <int R, int MAXW = CubeWaves<R>::WAVES, class F>
__device__ __forceinline__ void brick_apply_fn(const uint32_t *adj, const double *in, double *out, size_t step, uint32_t b, F f) {
  using TL = CubeTile<R>;
  __shared__ double lds[MAXW * TL::ELEMS];
  const int lane = threadIdx.x & 63, hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (threadIdx.x >> 6) * TL::ELEMS;
  b = __builtin_amdgcn_readfirstlane(b);
  cube_stage<R>(tile, adj, in, step, b, lane);
  const double *c = tile + TL::at(R, hi + R, lo + R);
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], f(c + k * TL::T * TL::P));
}

// ---------------------------------------------------------------------------------------------------------------
// STAR-shaped stencils given as data, radius up to 4: every tap moves along ONE axis (what a user-written radius-3 or
// radius-4 finite-difference script is; the reference's generator takes any script, codegen/st/codegen/base.py:678).  Staging
// as star_brick (k-column in registers, TJ / TI sub-tiles); with taps known only at run time the k-column goes to LDS as
// well (TK[8+2R][64]) so that a tap is ONE ds_read at base(axis) + offset + k * stride(axis).  General path, not a tuned one.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxStarTaps = 64;
struct StarTaps {
  int n;
  int axis[kMaxStarTaps]; // 0 = i, 1 = j, 2 = k (a centre tap is axis 0, offset 0)
  int off[kMaxStarTaps];  // -R .. R
  double w[kMaxStarTaps];
};

template <int R>
struct StarTapTile {
  using TL = StarTile<R>;
  static constexpr int TK = TL::ELEMS;                 // k-extended column planes behind the TJ / TI sub-tiles
  static constexpr int ELEMS = TL::ELEMS + (8 + 2 * R) * 64;
};

// loads + staging shared by the two star-tap forms; col[] = the lane's k-column, [R, R+8) its own planes
template <int R>
__device__ __forceinline__ void star_stage(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in, size_t step,
                                           uint32_t b, int lane, double (&col)[8 + 2 * R]) {
  using TL = StarTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  const uint32_t arow = adj[(size_t)b * 27 + (lane < 27 ? lane : 13)];
  const uint32_t n_km = __builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)), n_kp = __builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0));
  const uint32_t n_jm = __builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0)), n_jp = __builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0));
  const uint32_t n_im = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1)), n_ip = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1));
  const double *self = in + (size_t)b * step;
  const double *p_km = in + (size_t)n_km * step, *p_kp = in + (size_t)n_kp * step;
  const double *p_jm = in + (size_t)n_jm * step, *p_jp = in + (size_t)n_jp * step;
  const double *p_im = in + (size_t)n_im * step, *p_ip = in + (size_t)n_ip * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) col[R + k] = self[k * 64 + lane];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    col[r] = p_km[(8 - R + r) * 64 + lane];
    col[R + 8 + r] = p_kp[r * 64 + lane];
  }
  double fjm[R], fjp[R], fim[R], fip[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    fjm[r] = p_jm[hi * 64 + (8 - R + r) * 8 + lo];
    fjp[r] = p_jp[hi * 64 + r * 8 + lo];
    fim[r] = p_im[hi * 64 + lo * 8 + (8 - R + r)];
    fip[r] = p_ip[hi * 64 + lo * 8 + r];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    tile[TL::tj(k, hi + R, lo)] = col[R + k];
    tile[TL::ti(k, hi, lo + R)] = col[R + k];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tile[TL::tj(hi, r, lo)] = fjm[r];
    tile[TL::tj(hi, R + 8 + r, lo)] = fjp[r];
    tile[TL::ti(hi, lo, r)] = fim[r];
    tile[TL::ti(hi, lo, R + 8 + r)] = fip[r];
  }
}

// taps as run-time data (brk_stencil_apply_taps for star-shaped tables of radius 3 or 4)
template <int R>
__device__ __forceinline__ void star_taps_brick(double *tile, const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                double *__restrict__ out, size_t step, uint32_t b, const StarTaps &tp, int lane) {
  using TL = StarTile<R>;
  using TT = StarTapTile<R>;
  const int hi = lane >> 3, lo = lane & 7;
  double col[8 + 2 * R];
  star_stage<R>(tile, adj, in, step, b, lane, col);
#pragma unroll
  for (int q = 0; q < 8 + 2 * R; ++q) tile[TT::TK + q * 64 + lane] = col[q];
  wave_lds_sync();
  // per axis: address of the lane's centre cell in plane 0 and the stride from one plane to the next
  const int base[3] = {TL::ti(0, hi, lo + R), TL::tj(0, hi + R, lo), TT::TK + R * 64 + lane};
  const int tapstride[3] = {1, 8, 64};
  const int kstride[3] = {8 * TL::PI, TL::T * 8, 64};
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int t = 0; t < tp.n; ++t) {
    const int a = tp.axis[t];
    const int o = (a == 0 ? base[0] : (a == 1 ? base[1] : base[2])) + tp.off[t] * (a == 0 ? tapstride[0] : (a == 1 ? tapstride[1] : tapstride[2]));
    const int ks = a == 0 ? kstride[0] : (a == 1 ? kstride[1] : kstride[2]);
    const double w = tp.w[t];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += w * tile[o + k * ks];
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}

// the same with the tap table known at compile time (what tools/brkscatter emits for a star-shaped script of radius 3 or 4):
// fully unrolled, the k-taps straight from the register column
template <int R, int NTAPS, int MAXW = 4>
__device__ __forceinline__ void brick_apply_star_taps_static(const uint32_t *adj, const double *in, double *out, size_t step, uint32_t b,
                                                             const int (&axis)[NTAPS], const int (&off)[NTAPS], const double (&w)[NTAPS]) {
  using TL = StarTile<R>;
  __shared__ double lds[MAXW * TL::ELEMS];
  const int lane = threadIdx.x & 63, hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (threadIdx.x >> 6) * TL::ELEMS;
  b = __builtin_amdgcn_readfirstlane(b);
  double col[8 + 2 * R];
  star_stage<R>(tile, adj, in, step, b, lane, col);
  wave_lds_sync();
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const double v = axis[t] == 2 ? col[R + k + off[t]] : (axis[t] == 1 ? tile[TL::tj(k, hi + R + off[t], lo)] : tile[TL::ti(k, hi, lo + R + off[t])]);
      acc += w[t] * v;
    }
    store_result(&dst[k * 64 + lane], acc);
  }
}

// waves of one workgroup that fit the 64 KiB of static LDS with a StarTile<R> each
template <int R>
struct StarWaves {
  static constexpr int WAVES = R <= 2 ? 4 : 2;
};

// A STAR-shaped expression known at compile time (every grid reference moves along one axis: finite-difference scripts with limiters,
// the shape of stencils/cond.py): the star staging -- own brick and its six faces, no edge or corner gathers -- and f(ti, tj, ck) per
// cell, where in[p + (d, 0, 0)] is ti[d], in[p + (0, d, 0)] is tj[8 * d] and in[p + (0, 0, d)] is ck[d] (the lane's k-column, in registers).
// What brk_expr_compile and tools/brkscatter emit for such scripts instead of brick_apply_fn.
template <int R, int MAXW = StarWaves<R>::WAVES, class F>
__device__ __forceinline__ void brick_apply_star_fn(const uint32_t *adj, const double *in, double *out, size_t step, uint32_t b, F f) {
  using TL = StarTile<R>;
  static_assert(MAXW * TL::ELEMS * 8 <= 65536, "tiles exceed the static LDS of a workgroup");
  __shared__ double lds[MAXW * TL::ELEMS];
  const int lane = threadIdx.x & 63, hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (threadIdx.x >> 6) * TL::ELEMS;
  b = __builtin_amdgcn_readfirstlane(b);
  double col[8 + 2 * R];
  star_stage<R>(tile, adj, in, step, b, lane, col);
  wave_lds_sync();
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], f(tile + TL::ti(k, hi, lo + R), tile + TL::tj(k, hi + R, lo), &col[R + k]));
}

// ---------------------------------------------------------------------------------------------------------------
// Entry points for user kernels (one wave per brick, any number of waves per block up to MAXW).
// in/out: field base pointers (Brick::dat), step: Brick::step, adj: BrickInfo::adj as uint32_t[nbricks][27],
// coeff: DEVICE pointer to the coefficient vector of the stencil (see brk.h), b: wave-uniform brick id.
// ---------------------------------------------------------------------------------------------------------------
template <int STENCIL, int MAXW = 4>
__device__ __forceinline__ void brick_apply(const uint32_t *adj, const double *in, double *out, size_t step, uint32_t b,
                                            const double *coeff) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  b = __builtin_amdgcn_readfirstlane(b);
  Coeffs cf;
  if (coeff == nullptr) { // MPI_* constants of the reference's stencils/fake.h:11-33
    constexpr double dflt[BRK_NSTENCIL][10] = {{0}, {0}, {0.4, 0.1}, {0.4, 0.07, 0.03}, {0.1, 0.06, 0.045, 0.03, 0.015},
                                               {0.1, 0.04, 0.03, 0.01, 0.006, 0.004, 0.005, 0.002, 0.003, 0.001}, {0}};
    for (int s = 0; s < 10; ++s) cf.c[s] = dflt[STENCIL][s];
  } else if (STENCIL == BRK_27PT_COEFF) {
    for (int s = 0; s < 27; ++s) cf.c[s] = coeff[s];
    cf.c[adj_slot(0, 1, 0)] = 0.0;                 // reference quirk, stencils/3axis.cpp:195,225
    cf.c[adj_slot(1, 1, 0)] = coeff[16] + coeff[25];
  } else {
    constexpr int nc = (STENCIL == BRK_7PT_COEFF || STENCIL == BRK_COND7) ? 7 : (STENCIL == BRK_MPI7 ? 2 : (STENCIL == BRK_MPI13 ? 3 : (STENCIL == BRK_MPI25 ? 5 : 10)));
    for (int s = 0; s < nc; ++s) cf.c[s] = coeff[s];
  }
  if (STENCIL == BRK_27PT_COEFF || STENCIL == BRK_MPI125) {
    constexpr int KIND = STENCIL == BRK_27PT_COEFF ? kCube27 : kCube125;
    __shared__ double lds[MAXW * CubeTile<CubeTraits<KIND>::R>::ELEMS];
    cube_brick<KIND>(lds + wave * CubeTile<CubeTraits<KIND>::R>::ELEMS, adj, in, out, step, b, cf, lane);
  } else {
    constexpr int KIND = STENCIL == BRK_7PT_COEFF ? kStar7Coeff : (STENCIL == BRK_COND7 ? kStarCond7 : (STENCIL == BRK_MPI7 ? kStarMpi7 : (STENCIL == BRK_MPI13 ? kStarMpi13 : kStarMpi25)));
    __shared__ double lds[MAXW * StarTile<StarTraits<KIND>::R>::ELEMS];
    star_brick<KIND>(lds + wave * StarTile<StarTraits<KIND>::R>::ELEMS, adj, in, out, step, b, cf, lane);
  }
}

} // namespace brklib
#endif // BRICK_DEVICE_H
