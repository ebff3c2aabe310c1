// brk_cubedense_kernel.h -- the device side of the cube kernel k_cube_dense (brk_cubedense.hip includes it; brk_jit.hip carries it as
// text so that brk_expr_compile can instantiate the 27-point kernel's machinery -- dense tiles, sibling / fallback slots, exact links --
// around a user-written cell update: the unit defines brklib::jit::brk_user_cube and instantiates k_cube_dense<kCubeUser1, 4, 4, 1> or,
// for radius 2, <kCubeUser2, 4, 3, 5>).
// Device code only.
#ifndef BRK_CUBEDENSE_KERNEL_H
#define BRK_CUBEDENSE_KERNEL_H
#include "brick-device.h"
#include "brk_sbase.h"

namespace brklib {
#ifdef BRK_STAMP
// diagnostic build only (make EXTRA_HIPFLAGS=-DBRK_STAMP): per-phase shader-clock totals of k_cube_dense, summed over waves
constexpr unsigned kStampSlots = 1u << 19;
__device__ unsigned long long g_stamp[kStampSlots][6]; // one row per list slot: no contention
#define STAMP(var)                                                                          \
  unsigned long long var;                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                        \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");              \
  __builtin_amdgcn_sched_barrier(0)
extern "C" int brk_debug_stamps(unsigned long long out[8], int reset) {
  static unsigned long long *host = new unsigned long long[(size_t)kStampSlots * 6];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * kStampSlots * 6) != hipSuccess) return -1;
  for (int q = 0; q < 8; ++q) out[q] = 0;
  for (size_t sl = 0; sl < kStampSlots; ++sl)
    if (host[sl * 6 + 1]) {
      for (int q = 0; q < 6; ++q) out[q] += host[sl * 6 + q];
      ++out[7];
    }
  if (reset) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamp)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * kStampSlots * 6) != hipSuccess) return -1;
  }
  return 0;
}
// the same totals by position of the wave in its group of four (slot & 3): are the group's end waves the late ones?
extern "C" int brk_debug_stamps_by_wave(unsigned long long out[4][8]) {
  static unsigned long long *host = new unsigned long long[(size_t)kStampSlots * 6];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * kStampSlots * 6) != hipSuccess) return -1;
  for (int w = 0; w < 4; ++w)
    for (int q = 0; q < 8; ++q) out[w][q] = 0;
  for (size_t sl = 0; sl < kStampSlots; ++sl)
    if (host[sl * 6 + 1]) {
      for (int q = 0; q < 6; ++q) out[sl & 3][q] += host[sl * 6 + q];
      ++out[sl & 3][7];
    }
  return 0;
}
#else
#define STAMP(var)
#endif
// libbrk.so keeps the kernel in an anonymous namespace; a run-time compiled unit (BRK_JIT_CUBE) needs a name it can ask for
#ifdef BRK_JIT_CUBE
#define BRK_CD_NS jit
#else
#define BRK_CD_NS
#endif
namespace BRK_CD_NS {
#ifdef BRK_JIT_CUBE
// run-time compiled units (brk_expr_compile, programs of radius 1 or 2 that are not star-shaped) define the cell update themselves:
// in[p + (di, dj, dk)] = c[dk * (8 + 2R) * 8 + dj * 8 + col[di]] (c: the cell's row start in its plane of the dense tile, col: per-lane column
// offsets -- own tile, sibling tile or the slot on that side)
__device__ __forceinline__ double brk_user_cube(const double *c, const int *col);
#endif

// ablation experiments (diagnostic builds only, make EXTRA_HIPFLAGS=-DBRK_DIAG, tools/kbench -D bits; WRONG results on purpose):
//   1 no accumulation (the centre value is stored)   2 no edge/corner gathers   4 no k-/j-face loads   8 neighbours taken as linked
//   by position in the group (the rows of the list neighbours are not read)   16 no i-side loads at group ends   32 own brick not loaded
#ifdef BRK_DIAG
#define DG(x) ((diag & (x)) != 0)
#else
#define DG(x) false
#endif

template <int KIND>
struct DenseTile {
  static constexpr int R = CubeTraits<KIND>::R;
  static constexpr int T = 8 + 2 * R;
  // planes of the halo'd brick kept in LDS: all T of them, except for the diamond, whose outermost levels stay in registers
  static constexpr int KK0 = KIND == kDiamond25 ? 1 : 0, NPL = KIND == kDiamond25 ? T - 2 : T;
  static constexpr int ELEMS = NPL * T * 8;
  static_assert(ELEMS % 32 == 0, "sibling tiles must keep the bank pattern");
  __host__ __device__ static constexpr int at(int kk, int jj, int c) { return ((kk - KK0) * T + jj) * 8 + c; }
  __host__ __device__ static constexpr bool has_plane(int kk) { return kk >= KK0 && kk < KK0 + NPL; }
};

// One descriptor per cell:
//   offset inside the destination slot | adjacency slot << 16 | element offset inside the source brick << 21
// Group 0 (di = 0) goes into the wave's own tile, group 1 (di = -1) into columns 8-R..7 of the slot on the left, group 2
// (di = +1) into columns 0..R-1 of the slot on the right; a wave whose i-neighbour is a sibling skips that group entirely.
// The full cubes need every edge/corner cell of the halo; the 25-point diamond (two fused 7-point sweeps) only the twelve
// one-cell-wide edge lines next to the brick.
template <int KIND>
struct DX {
  static constexpr int R = CubeTraits<KIND>::R;
  static constexpr int T = 8 + 2 * R;
  __host__ __device__ static constexpr bool wanted(int kk, int jj, int ii) {
    const int dk = kk < R ? -1 : (kk >= R + 8 ? 1 : 0), dj = jj < R ? -1 : (jj >= R + 8 ? 1 : 0);
    const int di = ii < R ? -1 : (ii >= R + 8 ? 1 : 0);
    if ((dk != 0) + (dj != 0) + (di != 0) < 2) return false; // own brick and faces are staged by structured loads
    if (KIND != kDiamond25) return true;
    // distance of the cell from the brick along each axis (0 inside): the diamond reaches L1 distance 2
    const int ek = kk < R ? R - kk : (kk >= R + 8 ? kk - R - 7 : 0), ej = jj < R ? R - jj : (jj >= R + 8 ? jj - R - 7 : 0);
    const int ei = ii < R ? R - ii : (ii >= R + 8 ? ii - R - 7 : 0);
    return ek + ej + ei <= 2;
  }
  // The full radius-2 cube gathers PAIRS of cells: its halo is two columns wide and the edge blocks are rows of eight, so every
  // wanted cell has its i-neighbour (ii ^ 1 counted from the tile's column 0 = ii with R even) wanted too -- 16 bytes per lane,
  // half the gather instructions, and the two cells of a 16-byte segment are no longer requested by different lanes.
  static constexpr bool PAIR = KIND == kCube125 || KIND == kCubeUser2;
  static constexpr int count(int side) {
    int n = 0;
    for (int kk = 0; kk < T; ++kk)
      for (int jj = 0; jj < T; ++jj)
        for (int ii = 0; ii < T; ++ii) {
          const int di = ii < R ? -1 : (ii >= R + 8 ? 1 : 0);
          if (PAIR && (ii & 1)) continue;
          if (wanted(kk, jj, ii) && (di < 0 ? 1 : (di > 0 ? 2 : 0)) == side) ++n;
        }
    return n;
  }
  static constexpr int IT0 = (count(0) + 63) / 64, ITS = (count(1) + 63) / 64;
  static constexpr int EXTRA_ITERS = IT0 + 2 * ITS;
  // table layout: four descriptors per lane side by side, so that one 16-byte load brings descriptors 4g .. 4g+3 of a lane
  static constexpr int QUADS = (EXTRA_ITERS + 3) / 4;
  __host__ __device__ static constexpr int slot_of(int n) { return ((n / 64 / 4) * 64 + n % 64) * 4 + (n / 64) % 4; }
};

template <int KIND>
struct DenseExtra {
  alignas(16) uint32_t e[DX<KIND>::QUADS * 256];
};

template <int KIND>
constexpr DenseExtra<KIND> make_dense_extra() {
  DenseExtra<KIND> t{};
  using X = DX<KIND>;
  constexpr int R = X::R, T = X::T;
  using TL = DenseTile<KIND>;
  for (int q = 0; q < X::QUADS * 256; ++q) t.e[q] = 0xFFFFFFFFu;
  for (int side = 0; side < 3; ++side) {
    int n = (side == 0 ? 0 : (side == 1 ? X::IT0 : X::IT0 + X::ITS)) * 64;
    for (int kk = 0; kk < T; ++kk)
      for (int jj = 0; jj < T; ++jj)
        for (int ii = 0; ii < T; ++ii) {
          const int dk = kk < R ? -1 : (kk >= R + 8 ? 1 : 0);
          const int dj = jj < R ? -1 : (jj >= R + 8 ? 1 : 0);
          const int di = ii < R ? -1 : (ii >= R + 8 ? 1 : 0);
          if (!X::wanted(kk, jj, ii)) continue;
          if (X::PAIR && (ii & 1)) continue; // the odd cell travels with its even neighbour
          if ((di < 0 ? 1 : (di > 0 ? 2 : 0)) != side) continue;
          const int sk = (kk - R + 8) % 8, sj = (jj - R + 8) % 8, si = (ii - R + 8) % 8;
          const int col = di < 0 ? 8 - R + ii : (di > 0 ? ii - R - 8 : ii - R);
          t.e[X::slot_of(n++)] = (uint32_t)TL::at(kk, jj, col) | ((uint32_t)adj_slot(dk, dj, di) << 16) |
                     ((uint32_t)(sk * 64 + sj * 8 + si) << 21);
        }
  }
  return t;
}


__device__ const DenseExtra<kCube27> g_dense27 = make_dense_extra<kCube27>();
__device__ const DenseExtra<kCube125> g_dense125 = make_dense_extra<kCube125>();
#ifdef BRK_JIT_CUBE
template <int KIND>
__device__ __forceinline__ const uint32_t *dense_table() {
  static_assert(KIND == kCubeUser1 || KIND == kCubeUser2, "a run-time compiled unit instantiates the user kinds only");
  return KIND == kCubeUser1 ? g_dense27.e : g_dense125.e; // same cells, same slots as the 27- / 125-point stencils
}
#else
__device__ const DenseExtra<kDiamond25> g_dense_d25 = make_dense_extra<kDiamond25>();
template <int KIND>
__device__ __forceinline__ const uint32_t *dense_table() {
  if (KIND == kCube27) return g_dense27.e;
  if (KIND == kCube125) return g_dense125.e;
  return g_dense_d25.e;
}
#endif

// Do bricks `left` and `right` (adjacency rows in lanes 0..26 of rl / rr) see the same world across their common i-face?
// right(dk,dj,-1) == left(dk,dj,0) and right(dk,dj,0) == left(dk,dj,+1) for all 9 (dk,dj) -- this includes right.im == left
// and left.ip == right; entry (dk,dj,di) sits in lane (dk+1)*9+(dj+1)*3+(di+1).
__device__ __forceinline__ bool linked(uint32_t rl, uint32_t rr, int lane) {
  // left is one brick to the left of right, so left's entry in column di+1 must be right's entry in column di
  const uint32_t l_up = (uint32_t)__shfl((int)rl, lane + 1);
  const bool ok = !(lane < 27 && lane % 3 != 2) || rr == l_up;
  return __ballot(!ok) == 0ull;
}

template <int KIND, int NW, int OCC, int PIPE = 1>
__global__ __launch_bounds__(64 * NW, OCC) void k_cube_dense(
    const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out, size_t step,
    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf, int diag, const uint32_t *__restrict__ linkmask,
    const uint32_t *__restrict__ cubemask, uint32_t cs1, uint32_t cs2) {
  constexpr int R = CubeTraits<KIND>::R;
  using TL = DenseTile<KIND>;
  using X = DX<KIND>;
  constexpr int E = TL::ELEMS;
  __shared__ double lds[(NW + 1) * E];
  __shared__ uint32_t breaks[NW];
  __shared__ double jslot[KIND == kCube125 && PIPE == 5 ? NW * 192 : 1]; // output-side shift slots, one per wave
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
#ifdef BRK_STAMP
  // stamp builds, diagnostic bit 512: the hardware waves of a group take the list entries in REVERSE (hardware wave 0 = right end):
  // does a straggler at one end of the group follow the wave or the side?  (profiles/r05_c_*)
  const uint32_t hw_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t w = (diag & 512) ? NW - 1 - hw_wave : hw_wave;
#else
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
  const uint32_t slot = wg * NW + w;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (w + 1) * E;                         // slot 0 is F, tile v is slot v+1
  double *left = tile - E;                                  // tile w-1, or F for w = 0
  const int roff = w == NW - 1 ? -NW * E : E;               // tile w+1, or F for the last wave
  double *right = tile + roff;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };

  // The descriptors of the edge/corner gathers depend on the lane only.  They are requested before anything else: a load
  // issued after the brick's own loads returns after them (vector loads complete in order), and the gathers, which need
  // the descriptors for their addresses, would wait for a full round trip of the brick before they could even be issued.
  uint32_t xd[X::EXTRA_ITERS];
  {
    const uint4 *xt = reinterpret_cast<const uint4 *>(dense_table<KIND>());
#pragma unroll
    for (int g = 0; g < X::QUADS; ++g) {
      const uint4 q = xt[g * 64 + lane];
      if (4 * g < X::EXTRA_ITERS) xd[4 * g] = q.x;
      if (4 * g + 1 < X::EXTRA_ITERS) xd[4 * g + 1] = q.y;
      if (4 * g + 2 < X::EXTRA_ITERS) xd[4 * g + 2] = q.z;
      if (4 * g + 3 < X::EXTRA_ITERS) xd[4 * g + 3] = q.w;
    }
  }
  STAMP(t0);
  if (DG(128)) __builtin_amdgcn_s_setprio(3); // experiment: the fetch phase ahead, the compute phase behind
  const uint32_t b = slot < n ? __builtin_amdgcn_readfirstlane(id_of(slot)) : BRK_LIST_SKIP;
  const bool active = b != BRK_LIST_SKIP;
  bool link_l = false, link_r = false, brk_l = false;
  double own[8], fkm[R], fkp[R], fjm[R], fjp[R], fim[R], fip[R];
  double xv[X::EXTRA_ITERS];
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 xv2[X::EXTRA_ITERS];
  // R = 2: the two halo rows (columns) of a j- (i-) face are 16 contiguous bytes per lane -- ONE instruction per face, every
  // 128-byte line requested once (two 8-byte instructions ask for the same lines twice; the second request finds the miss of
  // the first still pending and the L1 stalls on that, see brk_starwg.hip)
  typedef const __attribute__((address_space(1))) d2 *gc2;
  constexpr bool W16 = R == 2;
  // the output-side 125-point form reads everything from LDS (no register copy of the brick's own column): the brick and its
  // k-faces are moved in 16-byte lanes too -- lane l of instruction q takes elements 2(l%32), +1 of plane 2q + l/32
  constexpr bool KF16 = (KIND == kCube125 || KIND == kCubeUser2) && PIPE == 5; // both planes of a k-face go to LDS: one full-wave instruction per side
  // The diamond as well since round 4, but only with the plane loop WITHOUT read-ahead (PIPE 0: 116 VGPRs): with the read-ahead the
  // kernel sits at its 128-register bound, hipcc spills two gathered values and waits for vmcnt(0) in the middle of the fetch phase to
  // store them (0.545 vs 0.430 ms).  8 instead of 15 vector memory instructions per wave: 0.427-0.430 vs 0.430-0.435 ms.
  constexpr bool LD16 = KF16 || KIND == kCube27 || KIND == kCubeUser1 || (KIND == kDiamond25 && PIPE == 0);
  // 27-point: ONE plane per side goes to LDS (plane 7 below -> level R-1, plane 0 above -> level R+8): both sides in one
  // instruction, half a wave each
  constexpr bool KF16H = LD16 && !KF16;
  d2 own2[4], km2 = {0.0, 0.0}, kp2 = {0.0, 0.0}, kh2 = {0.0, 0.0}, jh2 = {0.0, 0.0};
  const int l5 = lane & 31, lp = lane >> 5;
  d2 jm2 = {0.0, 0.0}, jp2 = {0.0, 0.0}, im2 = {0.0, 0.0}, ip2 = {0.0, 0.0};
  if (active) {
    const uint32_t b_prev = w > 0 ? __builtin_amdgcn_readfirstlane(id_of(slot - 1)) : BRK_LIST_SKIP;
    const uint32_t b_next = (w + 1 < NW && slot + 1 < n) ? __builtin_amdgcn_readfirstlane(id_of(slot + 1)) : BRK_LIST_SKIP;
    const int l27 = lane < 27 ? lane : 13;
#ifdef BRK_STAMP
    asm volatile("" ::"s"(b_prev), "s"(b_next));
#endif
    STAMP(t0a);
    // the three adjacency rows are requested BEFORE the brick itself: vector loads return in order, and everything
    // below (links, face addresses) waits for the rows -- behind the 4 KiB of the brick that wait is a full HBM round trip
    // (in-kernel stamps: 8500 of a wave's 15000 cycles went by before the face loads were issued)
    // lanes 0..26: the adjacency row.  Box entry point over a registered table: a brick whose 26 neighbours are the ids the box
    // numbering predicts (one bit per brick, k_regular_mask in its cube form; a scalar load) computes its row -- the vector load of
    // the row is the one dependent round trip through the saturated vector memory path in front of a wave's own requests
    uint32_t arow;
    const bool predicted = cubemask && ((cubemask[b >> 5] >> (b & 31u)) & 1u);
    if (predicted) arow = b + (uint32_t)(l27 / 9 - 1) * cs2 + (uint32_t)((l27 / 3) % 3 - 1) * cs1 + (uint32_t)(l27 % 3 - 1);
    else arow = adj[(size_t)b * 27 + l27];
    // registered adjacency table (brk_adjacency_register): two bits per brick say whether it is linked to its -i / +i adjacency
    // neighbour (k_link_mask, brk_kernels.hip), so the rows of the list neighbours are not read -- measured with them ablated
    // (profiles/r03_b_cube_kernel_ablations.txt): 3.7 % (27-point) and 4.4 % (125-point) of the sweep
    uint32_t lmw_l = 0, lmw_r = 0;
    if (linkmask) {
      const uint32_t *lw = linkmask + 4 * (size_t)(b >> 6) + ((b >> 4) & 2u);
      lmw_l = lw[0], lmw_r = lw[1];
    }
    uint32_t prow = 0, nrow = 0;
    if (!DG(8) && !linkmask) {
      prow = adj[(size_t)(b_prev != BRK_LIST_SKIP ? b_prev : b) * 27 + l27];
      nrow = adj[(size_t)(b_next != BRK_LIST_SKIP ? b_next : b) * 27 + l27];
    }
    __builtin_amdgcn_sched_barrier(0);
    const gcptr self = sbase(in + (size_t)b * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      own[k] = 0.5;
      if (!LD16 && !DG(32)) own[k] = self[k * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      own2[q] = d2{0.5, 0.25};
      if (LD16 && !DG(32)) own2[q] = *(gc2)(self + q * 128 + 2 * lane);
    }
    if (DG(8)) {
      link_l = b_prev != BRK_LIST_SKIP, link_r = b_next != BRK_LIST_SKIP;
    } else if (linkmask) {
      link_l = b_prev != BRK_LIST_SKIP && __builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1)) == b_prev && ((lmw_l >> (b & 31u)) & 1u);
      link_r = b_next != BRK_LIST_SKIP && __builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1)) == b_next && ((lmw_r >> (b & 31u)) & 1u);
    } else {
      link_l = b_prev != BRK_LIST_SKIP && linked(prow, arow, lane);
      link_r = b_next != BRK_LIST_SKIP && linked(arow, nrow, lane);
    }
    brk_l = b_prev != BRK_LIST_SKIP && !link_l; // two active neighbours that cannot share
    STAMP(t0b);
#ifdef BRK_STAMP
    if (lane == 0 && slot < kStampSlots) {
      g_stamp[slot][0] = t0a - t0;
      g_stamp[slot][1] = t0b - t0a;
      g_stamp[slot][2] = t0b; // completed below
    }
#endif
    const uint32_t n_im = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1));
    const uint32_t n_ip = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1));
    const gcptr p_km = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)) * step);
    const gcptr p_kp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0)) * step);
    const gcptr p_jm = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0)) * step);
    const gcptr p_jp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0)) * step);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      fkm[r] = fkp[r] = 0.0;
      if (!DG(4) && (!LD16 || (KF16H && !TL::has_plane(r)))) fkm[r] = p_km[(8 - R + r) * 64 + lane];
      if (!DG(4) && (!LD16 || (KF16H && !TL::has_plane(R + 8 + r)))) fkp[r] = p_kp[r * 64 + lane];
      fjm[r] = fjp[r] = fim[r] = fip[r] = 0.0;
      if (!W16 && !LD16 && !DG(4)) {
        fjm[r] = p_jm[hi * 64 + (8 - R + r) * 8 + lo];
        fjp[r] = p_jp[hi * 64 + r * 8 + lo];
      }
    }
    if (KF16 && !DG(4)) {
      km2 = *(gc2)(p_km + 6 * 64 + 2 * lane);
      kp2 = *(gc2)(p_kp + 2 * lane);
    }
    if (KF16H && !DG(4)) kh2 = *(gc2)((lp ? p_kp : p_km + 7 * 64) + 2 * l5);
    if (LD16 && R == 1 && !DG(4)) jh2 = *(gc2)((lp ? p_jp : p_jm + 56) + (l5 >> 2) * 64 + 2 * (l5 & 3)); // row 7 below / row 0 above of plane l5/4
    if (W16 && !DG(4)) { // lane (k = hi, chunk lo): rows 6, 7 / 0, 1 of plane k are 128 contiguous bytes
      jm2 = *(gc2)(p_jm + hi * 64 + 48 + 2 * lo);
      jp2 = *(gc2)(p_jp + hi * 64 + 2 * lo);
    }
    if (DG(16)) link_l = link_r = true; // from here on only the loads of the i-sides look at the links (the tiles' F slots stay unwritten)
    if (!link_l) {
      const gcptr p_im = sbase(in + (size_t)n_im * step);
      if (W16) im2 = *(gc2)(p_im + hi * 64 + lo * 8 + 6); // lane (k = hi, j = lo): columns 6, 7 of its row
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (!W16) fim[r] = p_im[hi * 64 + lo * 8 + (8 - R + r)];
    }
    if (!link_r) {
      const gcptr p_ip = sbase(in + (size_t)n_ip * step);
      if (W16) ip2 = *(gc2)(p_ip + hi * 64 + lo * 8);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (!W16) fip[r] = p_ip[hi * 64 + lo * 8 + r];
    }
    auto gather = [&](int t) {
      if (DG(2)) {
        xv[t] = 0.0, xv2[t] = d2{0.0, 0.0};
        return;
      }
      const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((xd[t] >> 16) & 31u) << 2), (int)arow);
      if (X::PAIR) xv2[t] = xd[t] != 0xFFFFFFFFu ? *(const d2 *)(in + (size_t)nb * step + ((xd[t] >> 21) & 511u)) : d2{0.0, 0.0};
      else xv[t] = xd[t] != 0xFFFFFFFFu ? in[(size_t)nb * step + ((xd[t] >> 21) & 511u)] : 0.0;
    };
#pragma unroll
    for (int t = 0; t < X::IT0; ++t) gather(t);
    if (!link_l) {
#pragma unroll
      for (int t = X::IT0; t < X::IT0 + X::ITS; ++t) gather(t);
    }
    if (!link_r) {
#pragma unroll
      for (int t = X::IT0 + X::ITS; t < X::EXTRA_ITERS; ++t) gather(t);
    }
  }
  STAMP(t1); // everything is requested
#ifdef BRK_STAMP
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): all loads back
#endif
  STAMP(t2);
  if (lane == 0) breaks[w] = brk_l ? 1u : 0u;
  __syncthreads();
  STAMP(t3); // first barrier passed
  // 2-colouring of the runs: phase of wave w = parity of the breaks at or before it
  const unsigned long long bm = __ballot(lane < NW && breaks[lane < NW ? lane : 0] != 0u);
  const int my_phase = __builtin_popcountll(bm & ((2ull << w) - 1ull)) & 1;
  const int nphase = bm ? 2 : 1;

  gptr dst = (gptr)(out + (size_t)(active ? b : 0) * step);
  asm volatile("" : "+s"(dst));
  for (int ph = 0; ph < nphase; ++ph) {
    const bool mine = active && my_phase == ph;
    if (mine) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (!LD16) tile[TL::at(k + R, hi + R, lo)] = own[k];
      if (LD16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *(d2 *)&tile[TL::at(2 * q + lp + R, (l5 >> 2) + R, 2 * (l5 & 3))] = own2[q];
        if (KF16) {
          *(d2 *)&tile[TL::at(lp, (l5 >> 2) + R, 2 * (l5 & 3))] = km2;
          *(d2 *)&tile[TL::at(R + 8 + lp, (l5 >> 2) + R, 2 * (l5 & 3))] = kp2;
        } else {
          *(d2 *)&tile[TL::at(lp ? R + 8 : R - 1, (l5 >> 2) + R, 2 * (l5 & 3))] = kh2;
          if (R == 1) *(d2 *)&tile[TL::at((l5 >> 2) + R, lp ? R + 8 : 0, 2 * (l5 & 3))] = jh2;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!LD16 && TL::has_plane(r)) tile[TL::at(r, hi + R, lo)] = fkm[r];
        if (!LD16 && TL::has_plane(R + 8 + r)) tile[TL::at(R + 8 + r, hi + R, lo)] = fkp[r];
        if (!W16 && !LD16) {
          tile[TL::at(hi + R, r, lo)] = fjm[r];
          tile[TL::at(hi + R, R + 8 + r, lo)] = fjp[r];
        }
      }
      if (W16) { // rows 0, 1 (R+8, R+9) of plane k are 16 contiguous doubles of the tile
        *(d2 *)&tile[TL::at(hi + R, 0, 0) + 2 * lo] = jm2;
        *(d2 *)&tile[TL::at(hi + R, R + 8, 0) + 2 * lo] = jp2;
      }
      if (!link_l) {
        if (W16) *(d2 *)&left[TL::at(hi + R, lo + R, 8 - R)] = im2;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (!W16) left[TL::at(hi + R, lo + R, 8 - R + r)] = fim[r];
      }
      if (!link_r) {
        if (W16) *(d2 *)&right[TL::at(hi + R, lo + R, 0)] = ip2;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (!W16) right[TL::at(hi + R, lo + R, r)] = fip[r];
      }
#pragma unroll
      for (int t = 0; t < X::IT0; ++t)
        if (xd[t] != 0xFFFFFFFFu) {
          if (X::PAIR) *(d2 *)&tile[xd[t] & 0xFFFFu] = xv2[t];
          else tile[xd[t] & 0xFFFFu] = xv[t];
        }
      if (!link_l) {
#pragma unroll
        for (int t = X::IT0; t < X::IT0 + X::ITS; ++t)
          if (xd[t] != 0xFFFFFFFFu) {
            if (X::PAIR) *(d2 *)&left[xd[t] & 0xFFFFu] = xv2[t];
            else left[xd[t] & 0xFFFFu] = xv[t];
          }
      }
      if (!link_r) {
#pragma unroll
        for (int t = X::IT0 + X::ITS; t < X::EXTRA_ITERS; ++t)
          if (xd[t] != 0xFFFFFFFFu) {
            if (X::PAIR) *(d2 *)&right[xd[t] & 0xFFFFu] = xv2[t];
            else right[xd[t] & 0xFFFFu] = xv[t];
          }
      }
    }
    STAMP(t4); // staged
    __syncthreads(); // the tiles of this phase are complete
    STAMP(t5);
#ifdef BRK_STAMP
    if (lane == 0 && active && ph == 0 && slot < kStampSlots) {
      g_stamp[slot][2] = t1 - g_stamp[slot][2];
      g_stamp[slot][3] = t2 - t1;
      g_stamp[slot][4] = t5 - t2;
    }
#endif
    if (mine) {
      if (DG(128)) __builtin_amdgcn_s_setprio(0);
      if (DG(64)) __builtin_amdgcn_s_setprio(3); // experiment: the compute phase ahead of the other waves of its SIMD
      // per-lane offset of column lo+di relative to the row start: own tile, or the slot on that side
      int col[2 * R + 1];
#pragma unroll
      for (int di = -R; di <= R; ++di) {
        const int c = lo + di;
        col[di + R] = c < 0 ? -E + c + 8 : (c > 7 ? roff + c - 8 : c);
      }
      if (DG(1)) { // ablation: no accumulation, the centre values leave as they came
#pragma unroll
        for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], tile[TL::has_plane(k + R) ? TL::at(k + R, hi + R, lo) : 0] + (double)col[0]);
      } else if (KIND == kDiamond25) {
        const int co[5] = {col[0], col[1], col[2], col[3], col[4]};
        diamond25_accumulate<TL::T * 8, 8, PIPE == 1>(tile + TL::at(0, hi + R, 0), co, cf, dst, lane, fkm[0], fkp[R - 1]);
      } else if (KIND == kCube125) {
        const int co[5] = {col[0], col[1], col[2], col[3], col[4]};
        if (PIPE == 1) cube125_accumulate<TL::T * 8, 8>(tile + (hi + R) * 8, co, cf, (double *)dst, lane);
        else if (PIPE == 2) cube125_accumulate_static<TL::T * 8, 8>(tile + (hi + R) * 8, co, cf, (double *)dst, lane);
        else if (PIPE == 3) cube125_accumulate_rows<TL::T * 8, 8, 3>(tile + (hi + R) * 8, co, cf, (double *)dst, lane);
        else if (PIPE == 4) cube125_accumulate_rows<TL::T * 8, 8, 5>(tile + (hi + R) * 8, co, cf, (double *)dst, lane);
        else if (PIPE == 5) cube125_accumulate_jout<TL::T * 8, 8>(tile, co, cf, jslot + (PIPE == 5 ? w * 192 : 0), (double *)dst, lane);
        else cube125_accumulate_simple<TL::T * 8, 8>(tile + (hi + R) * 8, co, cf, (double *)dst, lane);
#ifdef BRK_JIT_CUBE
      } else if (KIND == kCubeUser1 || KIND == kCubeUser2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], brk_user_cube(tile + (k + R) * TL::T * 8 + (hi + R) * 8, col + R));
#endif
      } else {
        double acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.0;
        const int rowbase = (hi + R) * 8;
#pragma unroll
        for (int kk = 0; kk < TL::T; ++kk) {
#pragma unroll
          for (int dj = -R; dj <= R; ++dj)
#pragma unroll
            for (int di = -R; di <= R; ++di) {
              const double v = tile[kk * TL::T * 8 + dj * 8 + rowbase + col[di + R]];
#pragma unroll
              for (int dk = -R; dk <= R; ++dk) {
                const int k = kk - R - dk; // output plane fed through tap dk
                if (k >= 0 && k < 8) acc[k] += cf.c[adj_slot(dk, dj, di)] * v;
              }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
      }
    }
    STAMP(t6);
#ifdef BRK_STAMP
    if (lane == 0 && active && ph == 0 && slot < kStampSlots) g_stamp[slot][5] = t6 - t5;
#endif
    if (ph + 1 < nphase) __syncthreads(); // phase 1 overwrites slots phase 0 was reading
  }
}

} // namespace
} // namespace brklib
#endif
