// brk_starwg_kernel.h -- the device side of the radius-2 / radius-4 star kernel k_star_wg (brk_starwg.hip includes it; brk_jit.hip carries
// it as text so that brk_expr_compile can instantiate the same kernel around a user-written cell update: the unit defines
// brklib::jit::brk_user_star and instantiates k_star_wg<kStarUser2 | kStarUser4, 4, OCC>).  Device code only.
#ifndef BRK_STARWG_KERNEL_H
#define BRK_STARWG_KERNEL_H
#include "brick-device.h"
#include "brk_sbase.h"

// libbrk.so keeps the kernel in an anonymous namespace; a run-time compiled unit (BRK_JIT_STAR) needs a name it can ask for
#ifdef BRK_JIT_STAR
#define BRK_SW_NS jit
#else
#define BRK_SW_NS
#endif
namespace brklib {
namespace BRK_SW_NS {
#ifdef BRK_JIT_STAR
// run-time compiled units (brk_expr_compile, star-shaped programs of radius 2..4) define the cell update themselves: ck = the lane's
// k-column around the cell (ck[d] = in[p + (0, 0, d)]), jm[r - 1] / jp[r - 1] = in[p + (0, -r, 0)] / (0, +r, 0), im / ip likewise along i
__device__ __forceinline__ double brk_user_star(const double *ck, const double *jm, const double *jp, const double *im, const double *ip);
#endif

template <int R>
struct WgTile {
  static constexpr int ROWS = 8 + 2 * R;
  static constexpr int IM = ROWS * 8;          // fallback slots [R][8] of the -i columns
  static constexpr int IP = IM + R * 8;        // fallback slots [R][8] of the +i columns
  static constexpr int PLANE = IP + R * 8;     // doubles per k-plane (R=2: 128, R=4: 192; multiples of 32 -> planes bank-aligned)
  static constexpr int TILE = 8 * PLANE;
};

#ifdef BRK_STAMP_SW
// diagnostic build only (make EXTRA_HIPFLAGS=-DBRK_STAMP_SW): s_memtime at the phase boundaries of k_star_wg, one row per slot
} // namespace
constexpr unsigned kStampSlotsW = 1u << 19;
__device__ unsigned long long g_stampw[kStampSlotsW][6];
#define STAMPW(var)                                                            \
  unsigned long long var;                                                      \
  __builtin_amdgcn_sched_barrier(0);                                           \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
  __builtin_amdgcn_sched_barrier(0)
extern "C" int brk_debug_stamps(unsigned long long out[8], int reset) {
  static unsigned long long *host = new unsigned long long[(size_t)kStampSlotsW * 6];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stampw), sizeof(unsigned long long) * kStampSlotsW * 6) != hipSuccess) return -1;
  for (int q = 0; q < 8; ++q) out[q] = 0;
  for (size_t sl = 0; sl < kStampSlotsW; ++sl)
    if (host[sl * 6 + 1]) {
      for (int q = 0; q < 6; ++q) out[q] += host[sl * 6 + q];
      ++out[7];
    }
  void *p = nullptr;
  if (reset && (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stampw)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * kStampSlotsW * 6) != hipSuccess)) return -1;
  return 0;
}
namespace {
#else
#define STAMPW(var)
#endif

template <int KIND, int NW, int OCC, int DIAG = 0>
__global__ __launch_bounds__(64 * NW, OCC) void k_star_wg(
    const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out, size_t step,
    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf, const uint32_t *__restrict__ faces) {
  constexpr int R = StarTraits<KIND>::R;
  using TL = WgTile<R>;
  __shared__ double lds[NW * TL::TILE];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wg * NW + w;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * TL::TILE;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };

  double col[8 + 2 * R];
  STAMPW(t0);
  const uint32_t b = slot < n ? __builtin_amdgcn_readfirstlane(id_of(slot)) : BRK_LIST_SKIP;
  const bool active = b != BRK_LIST_SKIP;
#ifdef BRK_STAMP_SW
  asm volatile("" ::"s"(b));
#endif
  STAMPW(t1); // id known
  bool share_im = false, share_ip = false;
  if (active) {
    // the brick's own planes need only its id: they are in flight before the adjacency row is even requested (scalar loads
    // return out of order, so every wait on them is lgkmcnt(0) -- anything issued behind such a wait starts a round trip late)
    const gcptr self = sbase(in + (size_t)b * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) col[R + k] = self[k * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    // registered adjacency: the six face ids come from one aligned 32-byte row of the compact table (k_face_table, brk_kernels.hip).
    // Two branches, not a selected base pointer: each path then reads through ONE restrict pointer with a uniform address and
    // stays a scalar load (a select between the two tables turned all six into vector loads, whose in-order completion made the
    // face loads wait for the brick's own planes).
    uint32_t n_km, n_kp, n_jm, n_jp, n_im, n_ip;
    if (faces) {
      const uint32_t *f = faces + (size_t)b * 8;
      n_km = __builtin_amdgcn_readfirstlane(f[0]), n_kp = __builtin_amdgcn_readfirstlane(f[1]);
      n_jm = __builtin_amdgcn_readfirstlane(f[2]), n_jp = __builtin_amdgcn_readfirstlane(f[3]);
      n_im = __builtin_amdgcn_readfirstlane(f[4]), n_ip = __builtin_amdgcn_readfirstlane(f[5]);
    } else {
      const uint32_t *row = adj + (size_t)b * 27;
      n_km = __builtin_amdgcn_readfirstlane(row[adj_slot(-1, 0, 0)]), n_kp = __builtin_amdgcn_readfirstlane(row[adj_slot(1, 0, 0)]);
      n_jm = __builtin_amdgcn_readfirstlane(row[adj_slot(0, -1, 0)]), n_jp = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 1, 0)]);
      n_im = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 0, -1)]), n_ip = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 0, 1)]);
    }
    const uint32_t b_prev = w > 0 ? __builtin_amdgcn_readfirstlane(id_of(slot - 1)) : 0xFFFFFFFFu;
    const uint32_t b_next = (w + 1 < NW && slot + 1 < n) ? __builtin_amdgcn_readfirstlane(id_of(slot + 1)) : 0xFFFFFFFFu;
    share_im = n_im == b_prev;
    share_ip = n_ip == b_next;
    STAMPW(t2); // adjacency + list neighbours known (own planes in flight)
    // DIAG 4 / 8: the k- / j-face loads address the brick itself (same instructions, lines already in the L1): timing experiments
    const gcptr p_km = sbase(in + (size_t)(DIAG & 4 ? b : n_km) * step), p_kp = sbase(in + (size_t)(DIAG & 4 ? b : n_kp) * step);
    const gcptr p_jm = sbase(in + (size_t)(DIAG & 8 ? b : n_jm) * step), p_jp = sbase(in + (size_t)(DIAG & 8 ? b : n_jp) * step);
    // Every 128-byte line is requested by ONE instruction of this wave.  A j-face is R rows of 64 B per k-plane = R/2 whole
    // lines: lane (k, c) takes 16 B of it per instruction (8 lanes = one line), R/2 instructions on disjoint lines.  The first
    // form of this kernel took one row per instruction, lane (k, i): rows 2q and 2q+1 share a line, so the second request hit a
    // line whose miss was still pending -- the L1 stalls on that (TCP_PENDING_STALL_CYCLES: 70 % of the kernel's cycles in
    // profiles/r02_c_pmc_27pt_list_mem.txt) and every other wave's requests wait behind it.
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) d2 *gc2;
    constexpr int H = R / 2;
    d2 fjm[H], fjp[H], fim[H], fip[H];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      col[r] = DIAG & 1 ? 0.25 * r : p_km[(8 - R + r) * 64 + lane]; // DIAG: timing experiments only (wrong results)
      col[R + 8 + r] = DIAG & 1 ? 0.5 * r : p_kp[r * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < H; ++q) {
      fjm[q] = DIAG & 2 ? d2{0.125 * q, 1.0} : *(gc2)(p_jm + hi * 64 + (8 - R) * 8 + q * 16 + 2 * lo); // rows 8-R+2q, 8-R+2q+1 of the -j brick, plane k = hi
      fjp[q] = DIAG & 2 ? d2{0.375 * q, 2.0} : *(gc2)(p_jp + hi * 64 + q * 16 + 2 * lo);               // rows 2q, 2q+1 of the +j brick
      fim[q] = fip[q] = d2{0.0, 0.0};
    }
    // i-faces (only without a sibling): the R halo columns of a (k, j) row are R x 8 contiguous bytes.  R = 2: lane (k, j) takes
    // the 16 B in one instruction.  R = 4: 32 B per row; instruction q takes the planes 4q..4q+3 with lane (k, j, half), so the
    // two instructions touch disjoint lines.
    const int ik = R == 2 ? hi : (lane >> 4), ij = R == 2 ? lo : ((lane >> 1) & 7), ih = R == 2 ? 0 : (lane & 1);
    if (!share_im) {
      const gcptr p_im = sbase(in + (size_t)n_im * step);
#pragma unroll
      for (int q = 0; q < H; ++q) fim[q] = *(gc2)(p_im + (ik + 4 * q) * 64 + ij * 8 + (8 - R) + 2 * ih);
    }
    if (!share_ip) {
      const gcptr p_ip = sbase(in + (size_t)n_ip * step);
#pragma unroll
      for (int q = 0; q < H; ++q) fip[q] = *(gc2)(p_ip + (ik + 4 * q) * 64 + ij * 8 + 2 * ih);
    }
    STAMPW(t3); // everything requested
#ifdef BRK_STAMP_SW
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    STAMPW(t4);
    if (lane == 0 && slot < kStampSlotsW) {
      g_stampw[slot][0] = t1 - t0;
      g_stampw[slot][1] = t2 - t1;
      g_stampw[slot][2] = t3 - t2;
      g_stampw[slot][3] = t4 - t3;
      g_stampw[slot][4] = t4;
    }
#endif
    // ---- stage
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[k * TL::PLANE + (hi + R) * 8 + lo] = col[R + k];
#pragma unroll
    for (int q = 0; q < H; ++q) { // rows 2q + lo/4, columns 2(lo%4), +1: 16 aligned bytes of the tile
      *(d2 *)&tile[hi * TL::PLANE + (2 * q) * 8 + 2 * lo] = fjm[q];
      *(d2 *)&tile[hi * TL::PLANE + (R + 8 + 2 * q) * 8 + 2 * lo] = fjp[q];
    }
    if (!share_im) {
#pragma unroll
      for (int q = 0; q < H; ++q) { // fallback slots [r][j] of plane k
        tile[(ik + 4 * q) * TL::PLANE + TL::IM + (2 * ih) * 8 + ij] = fim[q].x;
        tile[(ik + 4 * q) * TL::PLANE + TL::IM + (2 * ih + 1) * 8 + ij] = fim[q].y;
      }
    }
    if (!share_ip) {
#pragma unroll
      for (int q = 0; q < H; ++q) {
        tile[(ik + 4 * q) * TL::PLANE + TL::IP + (2 * ih) * 8 + ij] = fip[q].x;
        tile[(ik + 4 * q) * TL::PLANE + TL::IP + (2 * ih + 1) * 8 + ij] = fip[q].y;
      }
    }
  }
  __syncthreads(); // sibling tiles are readable
  if (!active) return;
  STAMPW(t5);
  gptr dst = (gptr)(out + (size_t)b * step);
  asm volatile("" : "+s"(dst));
  // per-lane plane-0 offsets of the 2R i-neighbours: own tile, sibling tile (one tile to the left/right, same bank as
  // the lane's own row) or the fallback slots
  int o_im[R], o_ip[R];
#pragma unroll
  for (int r = 1; r <= R; ++r) {
    const int cm = lo - r, cp = lo + r;
    o_im[r - 1] = cm >= 0 ? (hi + R) * 8 + cm : (share_im ? -TL::TILE + (hi + R) * 8 + cm + 8 : TL::IM + (cm + R) * 8 + hi);
    o_ip[r - 1] = cp < 8 ? (hi + R) * 8 + cp : (share_ip ? TL::TILE + (hi + R) * 8 + cp - 8 : TL::IP + (cp - 8) * 8 + hi);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double jm[R], jp[R], im[R], ip[R];
#pragma unroll
    for (int r = 1; r <= R; ++r) {
      jm[r - 1] = tile[k * TL::PLANE + (hi + R - r) * 8 + lo];
      jp[r - 1] = tile[k * TL::PLANE + (hi + R + r) * 8 + lo];
      im[r - 1] = tile[k * TL::PLANE + o_im[r - 1]];
      ip[r - 1] = tile[k * TL::PLANE + o_ip[r - 1]];
    }
#ifdef BRK_JIT_STAR
    if (KIND == kStarUser2 || KIND == kStarUser4) {
      store_result(&dst[k * 64 + lane], brk_user_star(&col[R + k], jm, jp, im, ip));
      continue;
    }
#endif
    // stencils/fake.h:41-53 (25-pt) / :73-79 (13-pt): outermost shell first, centre last
    double res = 0.0;
#pragma unroll
    for (int r = R; r >= 1; --r) {
      const double sh = col[R + k + r] + col[R + k - r] + jp[r - 1] + jm[r - 1] + ip[r - 1] + im[r - 1];
      res = (r == R) ? sh * cf.c[r] : res + sh * cf.c[r];
    }
    res = res + col[R + k] * cf.c[0];
    store_result(&dst[k * 64 + lane], res);
  }
  STAMPW(t6);
#ifdef BRK_STAMP_SW
  if (lane == 0 && slot < kStampSlotsW) {
    g_stampw[slot][4] = t5 - g_stampw[slot][4];
    g_stampw[slot][5] = t6 - t5;
  }
#endif
}

} // namespace
} // namespace brklib
#endif
