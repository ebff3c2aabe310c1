// brk_star7_kernel.h -- the device side of the 7-point kernel k_s7_wg (brk_star7.hip includes it; brk_jit.hip carries it as text so that
// brk_expr_compile can instantiate the same kernel around a user-written cell update: star-shaped scripts of radius 1 run on the
// headline kernel: the unit defines brklib::jit::brk_user_point7 and instantiates k_s7_wg<kUser, 8, mode>).  Device code only.
#ifndef BRK_STAR7_KERNEL_H
#define BRK_STAR7_KERNEL_H
#include "brick-device.h"
#include "brk_sbase.h"

// libbrk.so keeps the kernel in an anonymous namespace; a run-time compiled unit (BRK_JIT_POINT7) needs a name it can ask hiprtc for
#ifdef BRK_JIT_POINT7
#define BRK_S7_NS jit
#else
#define BRK_S7_NS
#endif
namespace brklib {
namespace BRK_S7_NS {

constexpr int kPlane = 96; // doubles per k-plane of the tile: 10 rows x 8 + 8 (i-) + 8 (i+)
constexpr int kTile = 8 * kPlane;

struct Faces {
  double own[8], km, kp, jm, jp, im, ip;
};

// Face-neighbour ids of brick b.  b is wave-uniform, so these are scalar (s_load) reads of the adjacency row:
// they are counted on lgkmcnt, not vmcnt, and therefore never entangle with the vector loads/stores in flight.
struct Nbr {
  uint32_t km, kp, jm, jp, im, ip;
};

// faces != nullptr: the compact table of a registered adjacency (k_face_table, brk_kernels.hip): one aligned 32-byte row per brick
__device__ __forceinline__ Nbr load_nbr(const uint32_t *__restrict__ adj, const uint32_t *__restrict__ faces, uint32_t b) {
  if (faces) {
    const uint32_t *f = faces + (size_t)b * 8;
    return Nbr{f[0], f[1], f[2], f[3], f[4], f[5]};
  }
  const uint32_t *row = adj + (size_t)b * 27;
  Nbr n;
  n.km = row[adj_slot(-1, 0, 0)];
  n.kp = row[adj_slot(1, 0, 0)];
  n.jm = row[adj_slot(0, -1, 0)];
  n.jp = row[adj_slot(0, 1, 0)];
  n.im = row[adj_slot(0, 0, -1)];
  n.ip = row[adj_slot(0, 0, 1)];
  return n;
}

__device__ __forceinline__ void stage(double *tile, const Faces &f, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[k * kPlane + (hi + 1) * 8 + lo] = f.own[k];
  tile[hi * kPlane + lo] = f.jm;
  tile[hi * kPlane + 72 + lo] = f.jp;
  tile[hi * kPlane + 80 + lo] = f.im;
  tile[hi * kPlane + 88 + lo] = f.ip;
}

template <int KIND>
__device__ __forceinline__ double point7(double kp, double km, double jp, double jm, double ip, double im, double c0,
                                         const Coeffs &cf);
// KIND: 0 = coefficient form (7pt.py), 1 = alpha/beta form (mpi7pt.py), 2 = d3cond (cond.py)
enum { kCoeff = 0, kMpi = 1, kCond = 2, kUser = 3 };
#ifdef BRK_JIT_POINT7
// run-time compiled units (brk_expr_compile, star-shaped programs of radius 1) define the cell update themselves
__device__ __forceinline__ double brk_user_point7(double kp, double km, double jp, double jm, double ip, double im, double c0);
#endif

template <int KIND>
__device__ __forceinline__ double point7(double kp, double km, double jp, double jm, double ip, double im, double c0,
                                         const Coeffs &cf) {
#ifdef BRK_JIT_POINT7
  if (KIND == kUser) return brk_user_point7(kp, km, jp, jm, ip, im, c0);
#endif
  if (KIND == kMpi) // stencils/fake.h:346-349
    return (kp + km + jp + jm + ip + im) * cf.c[1] + c0 * cf.c[0];
  if (KIND == kCond) { // stencils/3axis.cpp:104-111: taps clipped at zero, absolute value of the sum
    auto pos = [](double x) { return x > 0.0 ? x : 0.0; };
    const double p = cf.c[5] * pos(kp) + cf.c[6] * pos(km) + cf.c[3] * pos(jp) + cf.c[4] * pos(jm) + cf.c[1] * pos(ip) +
                     cf.c[2] * pos(im) + cf.c[0] * pos(c0);
    return p > 0 ? p : -p;
  }
  // stencils/3axis.cpp:60-63 term order
  return cf.c[5] * kp + cf.c[6] * km + cf.c[3] * jp + cf.c[4] * jm + cf.c[1] * ip + cf.c[2] * im + cf.c[0] * c0;
}


// ---- variant 5/6: workgroup of NW waves = NW consecutive list entries; i-faces shared through LDS ------------------
// Measured on MI355X (profiles/): the two i-face gathers (64 lanes x 8 B at 64-B stride, touching every line of the
// neighbour brick) cost more than everything else in the kernel; without them the sweep runs at the streaming
// ceiling.  When consecutive list entries are i-neighbours (the usual lexicographic order), the neighbour's data is
// already staged in the sibling wave's tile: the i-edge lanes just point their ds_read into that tile (same bank as
// the lane's own row -> still conflict-free).  Entries that are not i-neighbours fall back to the gather.
// Brick ids of a structured box inside a lexicographic numbering: id = base + (k*nj_tot + j)*ni_tot + i for
// (i,j,k) in [lo, lo+cnt).  The kernel walks the box in the order of bricklib_amd/order.py (one slab of j-rows per XCD,
// blocks of 8 rows swept over all k, i fastest) computed arithmetically, so no id list is read at all.
struct Box {
  uint32_t base, ni_tot, nj_tot;
  uint32_t lo[3], cnt[3];
};

enum { kModeRange = 0, kModeList = 1, kModeBox = 2 };

// (brick id, previous, next, active) of wave w of workgroup `g` of XCD `x` in box mode
__device__ __forceinline__ bool box_entry(const Box &bx, uint32_t x, uint32_t g, uint32_t w, uint32_t &b, bool &has_prev,
                                          bool &has_next, uint32_t *slab_edge = nullptr) {
  const uint32_t G = (bx.cnt[0] + 7) / 8, nk = bx.cnt[2];
  const uint32_t j_lo = (bx.cnt[1] * x) / 8, j_hi = (bx.cnt[1] * (x + 1)) / 8;
  const uint32_t rows_x = j_hi - j_lo, full = rows_x / 8, rem = rows_x % 8;
  const uint32_t per_full = nk * 8 * G;
  uint32_t blk, r, rows;
  if (g < full * per_full) {
    blk = g / per_full;
    r = g % per_full;
    rows = 8;
  } else {
    blk = full;
    r = g - full * per_full;
    rows = rem;
    if (rows == 0 || r >= nk * rows * G) return false;
  }
  const uint32_t k = r / (rows * G), r2 = r % (rows * G), j = r2 / G, ig = r2 % G;
  const uint32_t i = ig * 8 + w;
  if (i >= bx.cnt[0]) return false;
  b = bx.base + ((bx.lo[2] + k) * bx.nj_tot + (bx.lo[1] + j_lo + blk * 8 + j)) * bx.ni_tot + bx.lo[0] + i;
  if (slab_edge) *slab_edge = (blk * 8 + j == 0 ? 1u : 0u) | (blk * 8 + j + 1 == rows_x ? 2u : 0u); // row whose -j / +j neighbour row belongs to another XCD
  has_prev = w > 0;
  has_next = w + 1 < 8 && i + 1 < bx.cnt[0];
  return true;
}

#ifdef BRK_STAMP_S7
// diagnostic build only (make EXTRA_HIPFLAGS=-DBRK_STAMP_S7): s_memtime at the phase boundaries of k_s7_wg, one row per slot
} // namespace
constexpr unsigned kStampSlots7 = 1u << 19;
__device__ unsigned long long g_stamp7[kStampSlots7][6];
#define STAMP7(var)                                                            \
  unsigned long long var;                                                      \
  __builtin_amdgcn_sched_barrier(0);                                           \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
  __builtin_amdgcn_sched_barrier(0)
extern "C" int brk_debug_stamps(unsigned long long out[8], int reset) {
  static unsigned long long *host = new unsigned long long[(size_t)kStampSlots7 * 6];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp7), sizeof(unsigned long long) * kStampSlots7 * 6) != hipSuccess) return -1;
  for (int q = 0; q < 8; ++q) out[q] = 0;
  for (size_t sl = 0; sl < kStampSlots7; ++sl)
    if (host[sl * 6 + 1]) {
      for (int q = 0; q < 6; ++q) out[q] += host[sl * 6 + q];
      ++out[7];
    }
  void *p = nullptr;
  if (reset && (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamp7)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * kStampSlots7 * 6) != hipSuccess)) return -1;
  return 0;
}
namespace {
#else
#define STAMP7(var)
#endif

// load-policy experiments (make EXTRA_HIPFLAGS=-DBRK_NT_OWN / -DBRK_NT_LAST): non-temporal hint on the brick's own planes / on the
// faces whose lines are not needed again (the bricks below and behind are finished)
#ifdef BRK_NT_OWN
#define BRK_LD_OWN(p) __builtin_nontemporal_load(p)
#else
#define BRK_LD_OWN(p) (*(p))
#endif
#ifdef BRK_NT_LAST
#define BRK_LD_LAST(p) __builtin_nontemporal_load(p)
#else
#define BRK_LD_LAST(p) (*(p))
#endif
// DIAG (diagnostic builds only, make EXTRA_HIPFLAGS=-DBRK_DIAG, box mode; bits 2..8 give WRONG results on purpose -- what does each
// source of re-read traffic cost?): 1 = face neighbours by arithmetic on the lexicographic numbering, the adjacency row is not read;
// 2 = the j-faces of the two rows per XCD slab whose neighbour row belongs to another XCD are read from the brick itself;
// 4 = every j-face from the brick itself; 8 = every k-face from the brick itself
// IDPP (round-5 experiment, `make VARIANTS=1` builds only; the shipped instantiations have IDPP = 0): the +-i neighbours of the own plane
// come from the lane's neighbours by DPP (row_shr:1 / row_shl:1 on the two halves of the double) instead of two tile reads; only the
// lanes on an i-edge read the tile (ONE combined read per plane: lo == 0 from the left sibling / fallback slot, lo == 7 from the right).
template <int MPI, int NW, int MODE, int DIAG = 0, int IDPP = 0>
__global__ __launch_bounds__(64 * NW, (24 / NW) * NW / 4) void k_s7_wg(const uint32_t *__restrict__ adj,
                                                                      const double *__restrict__ in,
                                                                      double *__restrict__ out, size_t step,
                                                                      const uint32_t *__restrict__ ids, uint32_t first,
                                                                      uint32_t n, Box bx, Coeffs cf,
                                                                      const uint32_t *__restrict__ regmask,
                                                                      const uint32_t *__restrict__ faces) {
  __shared__ double lds[NW * kTile];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wg * NW + w;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * kTile;
  // A wave's life is mostly latency (it moves 8 KiB and waits), so the dependent chain in front of the first vector
  // load is kept short: the brick id and its two list neighbours come from three clamped scalar loads issued together
  // (no branches), the adjacency row is requested next, and the 8 loads of the brick itself -- which need only the id --
  // are issued before the adjacency row is waited for.
  STAMP7(t0);
  uint32_t b = 0, b_prev, b_next, diag_edge = 0;
  bool active, has_prev = w > 0, has_next = w + 1 < NW && slot + 1 < n;
  if (MODE == kModeBox) {
    static_assert(MODE != kModeBox || NW == 8, "box mode walks rows in groups of 8");
    uint32_t slab_edge = 0;
    active = box_entry(bx, blockIdx.x & 7u, wg_in_xcd(cf, per_xcd), w, b, has_prev, has_next, DIAG ? &slab_edge : nullptr);
    b = __builtin_amdgcn_readfirstlane(b);
    if (DIAG) diag_edge = __builtin_amdgcn_readfirstlane(slab_edge);
    b_prev = b - 1;
    b_next = b + 1;
  } else {
    const uint32_t sl = slot < n ? slot : n - 1;
    const uint32_t sp = sl > 0 ? sl - 1 : 0, sn = sl + 1 < n ? sl + 1 : n - 1;
    if (MODE == kModeList) {
      b = __builtin_amdgcn_readfirstlane(ids[sl]);
      b_prev = __builtin_amdgcn_readfirstlane(ids[sp]);
      b_next = __builtin_amdgcn_readfirstlane(ids[sn]);
    } else {
      b = first + sl;
      b_prev = first + sp;
      b_next = first + sn;
    }
    active = slot < n && b != BRK_LIST_SKIP; // padding entries keep groups of 8 aligned with rows of i-neighbours
  }
  Faces f;
  bool share_im = false, share_ip = false;
#ifdef BRK_STAMP_S7
  asm volatile("" ::"s"(b), "s"(b_prev), "s"(b_next));
#endif
  STAMP7(t1); // ids known
  if (active) {
    // Box mode over a REGISTERED adjacency table (brk_adjacency_register: the caller promised it does not change): bit b of
    // `regmask` says that the six face neighbours of brick b are the ones the lexicographic numbering predicts, so its adjacency
    // row is not read at all -- one bit instead of a 108-byte row that straddles two or three 64-byte lines (measured at 512^3,
    // diagnostic build: the rows cost 60 MB of fabric reads and 3.4 % of the sweep).  Other bricks read their row as always.
    constexpr bool kMaskable = MODE == kModeBox && !(DIAG & 1);
    const bool use_mask = kMaskable && regmask != nullptr;
    uint32_t mword = 0;
    if (use_mask) mword = (DIAG & 16) ? 0xFFFFFFFFu : regmask[b >> 5]; // DIAG 16: the mask path without its load (timing experiment)
    Nbr nb{};
    if (DIAG & 1) nb = Nbr{b - bx.ni_tot * bx.nj_tot, b + bx.ni_tot * bx.nj_tot, b - bx.ni_tot, b + bx.ni_tot, b - 1, b + 1};
    else if (!use_mask) nb = load_nbr(adj, faces, b);
    const gcptr self = sbase(in + (size_t)b * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) f.own[k] = BRK_LD_OWN(&self[k * 64 + lane]);
    __builtin_amdgcn_sched_barrier(0); // own loads are in flight before anything waits for the mask word / the adjacency row
    if (use_mask) {
      if ((mword >> (b & 31u)) & 1u) nb = Nbr{b - bx.ni_tot * bx.nj_tot, b + bx.ni_tot * bx.nj_tot, b - bx.ni_tot, b + bx.ni_tot, b - 1, b + 1};
      else nb = load_nbr(adj, faces, b);
    }
    if ((DIAG & 4) || ((DIAG & 2) && (diag_edge & 1u))) nb.jm = b;
    if ((DIAG & 4) || ((DIAG & 2) && (diag_edge & 2u))) nb.jp = b;
    if (DIAG & 8) nb.km = nb.kp = b;
    // siblings: entry slot-1 / slot+1 when they belong to this workgroup
    share_im = has_prev && __builtin_amdgcn_readfirstlane(nb.im) == b_prev;
    share_ip = has_next && __builtin_amdgcn_readfirstlane(nb.ip) == b_next;
    const gcptr p_km = sbase(in + (size_t)__builtin_amdgcn_readfirstlane(nb.km) * step);
    const gcptr p_kp = sbase(in + (size_t)__builtin_amdgcn_readfirstlane(nb.kp) * step);
    const gcptr p_jm = sbase(in + (size_t)__builtin_amdgcn_readfirstlane(nb.jm) * step);
    const gcptr p_jp = sbase(in + (size_t)__builtin_amdgcn_readfirstlane(nb.jp) * step);
    STAMP7(t2); // adjacency known (own loads in flight)
#ifdef BRK_STAMP_S7
    if (lane == 0 && slot < kStampSlots7) {
      g_stamp7[slot][0] = t1 - t0;
      g_stamp7[slot][1] = t2 - t1;
      g_stamp7[slot][2] = t2;
    }
#endif
    f.km = BRK_LD_LAST(&p_km[7 * 64 + lane]);
    f.kp = p_kp[lane];
    f.jm = BRK_LD_LAST(&p_jm[hi * 64 + 7 * 8 + lo]);
    f.jp = p_jp[hi * 64 + lo];
    f.im = 0.0;
    f.ip = 0.0;
    if (!share_im) f.im = sbase(in + (size_t)__builtin_amdgcn_readfirstlane(nb.im) * step)[hi * 64 + lo * 8 + 7];
    if (!share_ip) f.ip = sbase(in + (size_t)__builtin_amdgcn_readfirstlane(nb.ip) * step)[hi * 64 + lo * 8];
    STAMP7(t3); // everything requested
#ifdef BRK_STAMP_S7
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    STAMP7(t4);
    if (lane == 0 && slot < kStampSlots7) {
      g_stamp7[slot][2] = t3 - g_stamp7[slot][2];
      g_stamp7[slot][3] = t4 - t3;
      g_stamp7[slot][4] = t4;
    }
#endif
    stage(tile, f, lane);
  }
  __syncthreads(); // every wave's brick is staged; sibling tiles are now readable
  if (!active) return;
  STAMP7(t5);
  gptr dst = (gptr)(out + (size_t)b * step);
  asm volatile("" : "+s"(dst)); // see sbase()
  const int o_jm = hi * 8 + lo, o_jp = (hi + 2) * 8 + lo;
  const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : (share_im ? -kTile + (hi + 1) * 8 + 7 : 80 + hi);
  const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : (share_ip ? kTile + (hi + 1) * 8 : 88 + hi);
  const int o_edge = lo == 0 ? o_im : o_ip; // IDPP: the one tile read of an i-edge lane (interior lanes read a neighbour cell and drop it)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double jm = tile[k * kPlane + o_jm], jp = tile[k * kPlane + o_jp];
    const double c0 = f.own[k];
    double im, ip;
    if (IDPP) {
      const double edge = tile[k * kPlane + o_edge];
      const int c_lo = __double2loint(c0), c_hi = __double2hiint(c0);
      // row_shr:1 (0x111): lane L takes lane L - 1 of its 16-lane row; row_shl:1 (0x101): lane L + 1
      const double dm = __hiloint2double(__builtin_amdgcn_update_dpp(0, c_hi, 0x111, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, c_lo, 0x111, 0xF, 0xF, false));
      const double dp = __hiloint2double(__builtin_amdgcn_update_dpp(0, c_hi, 0x101, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, c_lo, 0x101, 0xF, 0xF, false));
      im = lo == 0 ? edge : dm;
      ip = lo == 7 ? edge : dp;
    } else {
      im = tile[k * kPlane + o_im], ip = tile[k * kPlane + o_ip];
    }
    const double km = k == 0 ? f.km : f.own[k > 0 ? k - 1 : 0];
    const double kp = k == 7 ? f.kp : f.own[k < 7 ? k + 1 : 7];
    double res;
    res = point7<MPI>(kp, km, jp, jm, ip, im, c0, cf);
    store_result(&dst[k * 64 + lane], res);
  }
  STAMP7(t6);
#ifdef BRK_STAMP_S7
  if (lane == 0 && slot < kStampSlots7) {
    g_stamp7[slot][4] = t5 - g_stamp7[slot][4];
    g_stamp7[slot][5] = t6 - t5;
  }
#endif
}

} // namespace
} // namespace brklib
#endif
