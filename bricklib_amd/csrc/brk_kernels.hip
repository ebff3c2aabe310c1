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
#include <cstring>

#include "brk_device.h"

using namespace brk;

namespace {

// Which entry of the brick list this wave owns (or n if none).  XCD-aware: see file header.
__device__ __forceinline__ uint32_t wave_slot(uint32_t n) {
  const uint32_t per_xcd = gridDim.x >> 3; // grid is a multiple of 8 blocks
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t w = wg * kWavesPerWG + (threadIdx.x >> 6);
  return w < n ? w : n;
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

enum StarKind { kStar7Coeff, kStarMpi7, kStarMpi13, kStarMpi25 };

template <int KIND>
struct StarTraits;
template <> struct StarTraits<kStar7Coeff> { static constexpr int R = 1; };
template <> struct StarTraits<kStarMpi7> { static constexpr int R = 1; };
template <> struct StarTraits<kStarMpi13> { static constexpr int R = 2; };
template <> struct StarTraits<kStarMpi25> { static constexpr int R = 4; };

template <int KIND>
__global__ __launch_bounds__(kWG) void k_star(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                               double *__restrict__ out, size_t step,
                                               const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                               Coeffs cf) {
  constexpr int R = StarTraits<KIND>::R;
  using TL = StarTile<R>;
  __shared__ double lds[kWavesPerWG * TL::ELEMS];

  const uint32_t slot = wave_slot(n);
  if (slot >= n) return; // wave-uniform; no workgroup barrier anywhere below
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7; // (j,i) when computing; (k,i) / (k,j) when staging faces
  double *tile = lds + (threadIdx.x >> 6) * TL::ELEMS;

  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
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
    dst[k * 64 + lane] = res;
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
  static constexpr int P = R == 1 ? 10 : 12; // i pitch in doubles (R=1 conflict-free; R=2 4 of 32 lanes 2-way)
  static constexpr int ELEMS = T * T * P;
  __host__ __device__ static constexpr int at(int kk, int jj, int ii) { return (kk * T + jj) * P + ii; }
  static constexpr int NEXTRA = T * T * T - 512 - 6 * 64 * R;
  static constexpr int EXTRA_ITERS = (NEXTRA + 63) / 64;
};

// descriptor = tile index | adjacency slot << 16 | element offset inside the source brick << 21
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
                   ((uint32_t)(sk * 64 + sj * 8 + si) << 21);
      }
  for (; n < CubeTile<R>::EXTRA_ITERS * 64; ++n) t.e[n] = 0xFFFFFFFFu;
  return t;
}

__device__ const ExtraTable<1> g_extra1 = make_extra_table<1>();
__device__ const ExtraTable<2> g_extra2 = make_extra_table<2>();

template <int R>
__device__ __forceinline__ const uint32_t *extra_table() {
  if (R == 1) return g_extra1.e;
  return g_extra2.e;
}

enum CubeKind { kCube27, kCube125 };
template <int KIND> struct CubeTraits;
template <> struct CubeTraits<kCube27> { static constexpr int R = 1; };
template <> struct CubeTraits<kCube125> { static constexpr int R = 2; };

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

template <int KIND>
__global__ __launch_bounds__(kWG) void k_cube(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                               double *__restrict__ out, size_t step,
                                               const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                               Coeffs cf) {
  constexpr int R = CubeTraits<KIND>::R;
  using TL = CubeTile<R>;
  __shared__ double lds[kWavesPerWG * TL::ELEMS];

  const uint32_t slot = wave_slot(n);
  if (slot >= n) return;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (threadIdx.x >> 6) * TL::ELEMS;

  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
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
  const uint32_t *xt = extra_table<R>();
  double xv[TL::EXTRA_ITERS];
  uint32_t xd[TL::EXTRA_ITERS];
#pragma unroll
  for (int t = 0; t < TL::EXTRA_ITERS; ++t) {
    xd[t] = xt[t * 64 + lane];
    const bool live = xd[t] != 0xFFFFFFFFu;
    // neighbour id of this lane's cell: pull lane (slot) of the adjacency-row register
    const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((xd[t] >> 16) & 31u) << 2), (int)arow);
    xv[t] = live ? in[(size_t)nb * step + (xd[t] >> 21)] : 0.0;
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

  // ---- scatter-form accumulate; lane = (j=hi, i=lo)
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
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
          if (k >= 0 && k < 8) {
            const double w = KIND == kCube27 ? cf.c[adj_slot(dk, dj, di)] : cf.c[cls125(dk, dj, di)];
            acc[k] += w * v;
          }
        }
      }
  }
  double *dst = out + (size_t)b * step;
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k * 64 + lane] = acc[k];
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

int g_variant[BRK_NSTENCIL] = {0, 0, 0, 0, 0, 0};
constexpr int kDefaultStar7Variant = 5; // 8-wave workgroups sharing i-faces through LDS (brk_star7.hip)

// defaults: stencils/fake.h:11-33
const double kDefault[BRK_NSTENCIL][10] = {{0}, {0}, {0.4, 0.1}, {0.4, 0.07, 0.03}, {0.1, 0.06, 0.045, 0.03, 0.015},
                                           {0.1, 0.04, 0.03, 0.01, 0.006, 0.004, 0.005, 0.002, 0.003, 0.001}};

int make_coeffs(int id, const double *host, Coeffs &cf) {
  std::memset(&cf, 0, sizeof cf);
  const int nc = brk_stencil_ncoeff(id);
  if (nc < 0) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
  if (!host) {
    if (id == BRK_7PT_COEFF || id == BRK_27PT_COEFF)
      return brk_fail(BRK_EINVAL, "stencil %d needs a coefficient vector", id);
    host = kDefault[id];
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

int launch_stencil(int id, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const double *coeff_host, brk_stream s) {
  if (id < 0 || id >= BRK_NSTENCIL) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
  if (!adj || !in || !out) return brk_fail(BRK_EINVAL, "null device pointer");
  if (in == out) return brk_fail(BRK_EINVAL, "in and out must be different fields");
  if (step < 512) return brk_fail(BRK_EINVAL, "step_elems %zu < 512", step);
  if (n == 0) return BRK_OK;
  Coeffs cf;
  if (int rc = make_coeffs(id, coeff_host, cf)) return rc;
  const uint32_t wgs = (n + kWavesPerWG - 1) / kWavesPerWG;
  const dim3 grid(((wgs + 7) / 8) * 8), block(kWG);
  hipStream_t st = static_cast<hipStream_t>(s);
  // 7-point sweeps: specialised kernels (brk_star7.hip) unless variant 1 (generic star kernel) is requested
  if ((id == BRK_7PT_COEFF || id == BRK_MPI7) && g_variant[id] != 1)
    return launch_star7(id, g_variant[id] == 0 ? kDefaultStar7Variant : g_variant[id], adj, in, out, step, ids, first, n, cf, st);
  switch (id) {
  case BRK_7PT_COEFF: hipLaunchKernelGGL(k_star<kStar7Coeff>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI7: hipLaunchKernelGGL(k_star<kStarMpi7>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI13: hipLaunchKernelGGL(k_star<kStarMpi13>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI25: hipLaunchKernelGGL(k_star<kStarMpi25>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_27PT_COEFF: hipLaunchKernelGGL(k_cube<kCube27>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  case BRK_MPI125: hipLaunchKernelGGL(k_cube<kCube125>, grid, block, 0, st, adj, in, out, step, ids, first, n, cf); break;
  }
  HIP_TRY(hipGetLastError());
  return BRK_OK;
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

extern "C" int brk_set_variant(int id, int v) {
  if (id < 0 || id >= BRK_NSTENCIL) return brk_fail(BRK_EINVAL, "unknown stencil id %d", id);
  g_variant[id] = v;
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
