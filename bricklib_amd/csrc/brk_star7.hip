// brk_star7.hip -- the headline kernel: 7-point fp64 sweep over 8x8x8 bricks on gfx950.
//
// One wave64 per brick, lane = (j,i), k walked in registers.  Per brick a wave moves 4 KiB in, 4 KiB out and
// ~3 KiB of neighbour faces (which should come from L2/Infinity Cache, not HBM).  The kernel is latency/occupancy
// bound, not ALU bound (56 fp64 FMAs per lane per brick), so everything here is about bytes in flight:
//   * LDS tile of 6 KiB per wave (24 waves/CU): per k-plane a dense 10x8 block (j-halo row, 8 centre rows,
//     j-halo row; one 32-lane half reads 32 consecutive doubles = conflict-free) followed by the 8+8 i-halo values;
//     lanes on an i edge simply point their ds_read at the halo slot, so every plane costs four ds_read_b64.
//   * persistent variant: a wave walks a strided sequence of bricks of its XCD's contiguous chunk and keeps a
//     3-deep software pipeline (brick id -> adjacency row -> 14 data loads) so that the dependent-load chain of
//     brick t+1/t+2 is in flight while brick t is staged, computed and stored.
#include "brk_device.h"

namespace brk {
namespace {

constexpr int kPlane = 96; // doubles per k-plane of the tile: 10 rows x 8 + 8 (i-) + 8 (i+)
constexpr int kTile = 8 * kPlane;

struct Faces {
  double own[8], km, kp, jm, jp, im, ip;
};

// Global-address-space pointers whose (wave-uniform) value is hidden from the optimiser: accesses through them
// are emitted as global_load/global_store with an SGPR base and a 32-bit per-lane offset.  Without this hipcc
// folds base+lane into 64-bit VGPR addresses (two VGPRs and two VALU ops per access).  The address space must be
// kept explicit: a generic pointer after the asm would turn the accesses into flat_* instructions, which force
// vmcnt(0)/lgkmcnt(0) waits.
typedef const __attribute__((address_space(1))) double *gcptr;
typedef __attribute__((address_space(1))) double *gptr;

__device__ __forceinline__ gcptr scalar_base(const double *p) {
  gcptr g = (gcptr)p;
  asm volatile("" : "+s"(g));
  return g;
}

// Face-neighbour ids of brick b.  b is wave-uniform, so these are scalar (s_load) reads of the adjacency row:
// they are counted on lgkmcnt, not vmcnt, and therefore never entangle with the vector loads/stores in flight.
struct Nbr {
  uint32_t km, kp, jm, jp, im, ip;
};

__device__ __forceinline__ Nbr load_nbr(const uint32_t *__restrict__ adj, uint32_t b) {
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

// SKIP (diagnostic variants 16..23 only, results are wrong): bit0 = no i-face loads, bit1 = no j-face loads,
// bit2 = no k-face loads; the skipped faces are fed from the brick's own data so that the rest of the kernel is unchanged.
template <int SKIP = 0>
__device__ __forceinline__ void issue_loads(Faces &f, const double *__restrict__ in, size_t step, uint32_t b,
                                            const Nbr &nb, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  const uint32_t n_km = __builtin_amdgcn_readfirstlane(nb.km), n_kp = __builtin_amdgcn_readfirstlane(nb.kp);
  const uint32_t n_jm = __builtin_amdgcn_readfirstlane(nb.jm), n_jp = __builtin_amdgcn_readfirstlane(nb.jp);
  const uint32_t n_im = __builtin_amdgcn_readfirstlane(nb.im), n_ip = __builtin_amdgcn_readfirstlane(nb.ip);
  const gcptr self = scalar_base(in + (size_t)b * step);
  const gcptr p_km = scalar_base(in + (size_t)n_km * step), p_kp = scalar_base(in + (size_t)n_kp * step);
  const gcptr p_jm = scalar_base(in + (size_t)n_jm * step), p_jp = scalar_base(in + (size_t)n_jp * step);
  const gcptr p_im = scalar_base(in + (size_t)n_im * step), p_ip = scalar_base(in + (size_t)n_ip * step);
#pragma unroll
  for (int k = 0; k < 8; ++k) f.own[k] = self[k * 64 + lane];
  f.km = (SKIP & 4) ? f.own[0] : p_km[7 * 64 + lane];          // plane k=7 of the brick below
  f.kp = (SKIP & 4) ? f.own[1] : p_kp[lane];                   // plane k=0 of the brick above
  f.jm = (SKIP & 2) ? f.own[2] : p_jm[hi * 64 + 7 * 8 + lo];   // lane=(k,i): row j=7 of the -j brick
  f.jp = (SKIP & 2) ? f.own[3] : p_jp[hi * 64 + lo];           // row j=0 of the +j brick
  f.im = (SKIP & 1) ? f.own[4] : p_im[hi * 64 + lo * 8 + 7];   // lane=(k,j): column i=7 of the -i brick
  f.ip = (SKIP & 1) ? f.own[5] : p_ip[hi * 64 + lo * 8];       // column i=0 of the +i brick
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

template <bool MPI>
__device__ __forceinline__ void compute_store(const double *tile, const Faces &f, double *dst_, int lane,
                                              const Coeffs &cf) {
  const int hi = lane >> 3, lo = lane & 7;
  gptr dst = (gptr)dst_;
  asm volatile("" : "+s"(dst)); // see scalar_base()
  // per-lane tile offsets of the four in-plane neighbours in plane 0 (plane k adds k*kPlane, an immediate)
  const int o_jm = hi * 8 + lo, o_jp = (hi + 2) * 8 + lo;
  const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : 80 + hi;
  const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : 88 + hi;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double jm = tile[k * kPlane + o_jm], jp = tile[k * kPlane + o_jp];
    const double im = tile[k * kPlane + o_im], ip = tile[k * kPlane + o_ip];
    const double c0 = f.own[k];
    const double km = k == 0 ? f.km : f.own[k > 0 ? k - 1 : 0];
    const double kp = k == 7 ? f.kp : f.own[k < 7 ? k + 1 : 7];
    double res;
    if (MPI) // stencils/fake.h:346-349
      res = (kp + km + jp + jm + ip + im) * cf.c[1] + c0 * cf.c[0];
    else     // stencils/3axis.cpp:60-63 term order
      res = cf.c[5] * kp + cf.c[6] * km + cf.c[3] * jp + cf.c[4] * jm + cf.c[1] * ip + cf.c[2] * im + cf.c[0] * c0;
    dst[k * 64 + lane] = res;
  }
}

// ---- variant 2: one brick per wave ------------------------------------------------------------------------------
template <bool MPI, int SKIP = 0>
__global__ __launch_bounds__(kWG, 6) void k_s7(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                double *__restrict__ out, size_t step,
                                                const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                                Coeffs cf) {
  __shared__ double lds[kWavesPerWG * kTile];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t slot = wg * kWavesPerWG + (threadIdx.x >> 6);
  if (slot >= n) return;
  const int lane = threadIdx.x & 63;
  double *tile = lds + (threadIdx.x >> 6) * kTile;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);
  Faces f;
  issue_loads<SKIP>(f, in, step, b, load_nbr(adj, b), lane);
  stage(tile, f, lane);
  wave_lds_sync();
  compute_store<MPI>(tile, f, out + (size_t)b * step, lane, cf);
}

// ---- variant 5/6: workgroup of NW waves = NW consecutive list entries; i-faces shared through LDS ------------------
// Measured on MI355X (profiles/): the two i-face gathers (64 lanes x 8 B at 64-B stride, touching every line of the
// neighbour brick) cost more than everything else in the kernel; without them the sweep runs at the streaming
// ceiling.  When consecutive list entries are i-neighbours (the usual lexicographic order), the neighbour's data is
// already staged in the sibling wave's tile: the i-edge lanes just point their ds_read into that tile (same bank as
// the lane's own row -> still conflict-free).  Entries that are not i-neighbours fall back to the gather.
template <bool MPI, int NW>
__global__ __launch_bounds__(64 * NW, (24 / NW) * NW / 4) void k_s7_wg(const uint32_t *__restrict__ adj,
                                                                      const double *__restrict__ in,
                                                                      double *__restrict__ out, size_t step,
                                                                      const uint32_t *__restrict__ ids, uint32_t first,
                                                                      uint32_t n, Coeffs cf) {
  __shared__ double lds[NW * kTile];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wg * NW + w;
  const bool active = slot < n;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * kTile;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };
  Faces f;
  uint32_t b = 0;
  bool share_im = false, share_ip = false;
  if (active) {
    b = __builtin_amdgcn_readfirstlane(id_of(slot));
    const Nbr nb = load_nbr(adj, b);
    // siblings: entry slot-1 / slot+1 when they belong to this workgroup
    const uint32_t b_prev = w > 0 ? __builtin_amdgcn_readfirstlane(id_of(slot - 1)) : 0xFFFFFFFFu;
    const uint32_t b_next = (w + 1 < NW && slot + 1 < n) ? __builtin_amdgcn_readfirstlane(id_of(slot + 1)) : 0xFFFFFFFFu;
    share_im = __builtin_amdgcn_readfirstlane(nb.im) == b_prev;
    share_ip = __builtin_amdgcn_readfirstlane(nb.ip) == b_next;
    const gcptr self = scalar_base(in + (size_t)b * step);
    const gcptr p_km = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb.km) * step);
    const gcptr p_kp = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb.kp) * step);
    const gcptr p_jm = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb.jm) * step);
    const gcptr p_jp = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb.jp) * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) f.own[k] = self[k * 64 + lane];
    f.km = p_km[7 * 64 + lane];
    f.kp = p_kp[lane];
    f.jm = p_jm[hi * 64 + 7 * 8 + lo];
    f.jp = p_jp[hi * 64 + lo];
    f.im = 0.0;
    f.ip = 0.0;
    if (!share_im) f.im = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb.im) * step)[hi * 64 + lo * 8 + 7];
    if (!share_ip) f.ip = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb.ip) * step)[hi * 64 + lo * 8];
    stage(tile, f, lane);
  }
  __syncthreads(); // every wave's brick is staged; sibling tiles are now readable
  if (!active) return;
  gptr dst = (gptr)(out + (size_t)b * step);
  asm volatile("" : "+s"(dst)); // see scalar_base()
  const int o_jm = hi * 8 + lo, o_jp = (hi + 2) * 8 + lo;
  const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : (share_im ? -kTile + (hi + 1) * 8 + 7 : 80 + hi);
  const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : (share_ip ? kTile + (hi + 1) * 8 : 88 + hi);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double jm = tile[k * kPlane + o_jm], jp = tile[k * kPlane + o_jp];
    const double im = tile[k * kPlane + o_im], ip = tile[k * kPlane + o_ip];
    const double c0 = f.own[k];
    const double km = k == 0 ? f.km : f.own[k > 0 ? k - 1 : 0];
    const double kp = k == 7 ? f.kp : f.own[k < 7 ? k + 1 : 7];
    double res;
    if (MPI)
      res = (kp + km + jp + jm + ip + im) * cf.c[1] + c0 * cf.c[0];
    else
      res = cf.c[5] * kp + cf.c[6] * km + cf.c[3] * jp + cf.c[4] * jm + cf.c[1] * ip + cf.c[2] * im + cf.c[0] * c0;
    dst[k * 64 + lane] = res;
  }
}

// ---- variant 7/8: as variant 5, and each wave owns KS bricks stacked in k -----------------------------------------
// List layout per workgroup: KS groups of 8 entries; group s+1 holds the +k neighbours of group s (checked at run
// time; any other list is still computed correctly, the k-faces are then loaded).  The k-face between two stacked
// bricks never leaves the register file, so k-face traffic drops by (KS-1)/KS; all loads of the KS bricks are issued
// up front.  The tile is reused brick after brick with two workgroup barriers per brick.
template <bool MPI, int KS>
__global__ __launch_bounds__(512, 4) void k_s7_wgk(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                    double *__restrict__ out, size_t step,
                                                    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                                    Coeffs cf) {
  constexpr int NW = 8;
  __shared__ double lds[NW * kTile];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * kTile;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };

  Faces f[KS];
  uint32_t b[KS];
  bool active[KS], share_im[KS], share_ip[KS], chain[KS]; // chain[s]: brick s+1 is the +k neighbour of brick s
  Nbr nb[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const uint32_t slot = (wg * KS + s) * NW + w;
    active[s] = slot < n;
    b[s] = 0;
    share_im[s] = share_ip[s] = chain[s] = false;
    if (active[s]) {
      b[s] = __builtin_amdgcn_readfirstlane(id_of(slot));
      nb[s] = load_nbr(adj, b[s]);
      const uint32_t b_prev = w > 0 ? __builtin_amdgcn_readfirstlane(id_of(slot - 1)) : 0xFFFFFFFFu;
      const uint32_t b_next = (w + 1 < NW && slot + 1 < n) ? __builtin_amdgcn_readfirstlane(id_of(slot + 1)) : 0xFFFFFFFFu;
      share_im[s] = __builtin_amdgcn_readfirstlane(nb[s].im) == b_prev;
      share_ip[s] = __builtin_amdgcn_readfirstlane(nb[s].ip) == b_next;
    }
  }
#pragma unroll
  for (int s = 0; s + 1 < KS; ++s)
    chain[s] = active[s] && active[s + 1] && __builtin_amdgcn_readfirstlane(nb[s].kp) == b[s + 1] &&
               __builtin_amdgcn_readfirstlane(nb[s + 1].km) == b[s];
  // ---- every global load of the KS bricks, up front
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (!active[s]) continue;
    const gcptr self = scalar_base(in + (size_t)b[s] * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[s].own[k] = self[k * 64 + lane];
    f[s].jm = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb[s].jm) * step)[hi * 64 + 7 * 8 + lo];
    f[s].jp = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb[s].jp) * step)[hi * 64 + lo];
    f[s].km = f[s].kp = f[s].im = f[s].ip = 0.0;
    if (!(s > 0 && chain[s > 0 ? s - 1 : 0]))
      f[s].km = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb[s].km) * step)[7 * 64 + lane];
    if (!(s + 1 < KS && chain[s]))
      f[s].kp = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb[s].kp) * step)[lane];
    if (!share_im[s]) f[s].im = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb[s].im) * step)[hi * 64 + lo * 8 + 7];
    if (!share_ip[s]) f[s].ip = scalar_base(in + (size_t)__builtin_amdgcn_readfirstlane(nb[s].ip) * step)[hi * 64 + lo * 8];
  }
  const int o_jm = hi * 8 + lo, o_jp = (hi + 2) * 8 + lo;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (active[s]) stage(tile, f[s], lane);
    __syncthreads(); // tiles of brick level s are staged; sibling tiles readable
    if (active[s]) {
      gptr dst = (gptr)(out + (size_t)b[s] * step);
      asm volatile("" : "+s"(dst)); // see scalar_base()
      const int o_im = lo > 0 ? (hi + 1) * 8 + lo - 1 : (share_im[s] ? -kTile + (hi + 1) * 8 + 7 : 80 + hi);
      const int o_ip = lo < 7 ? (hi + 1) * 8 + lo + 1 : (share_ip[s] ? kTile + (hi + 1) * 8 : 88 + hi);
      const bool up = s > 0 && chain[s > 0 ? s - 1 : 0], dn = s + 1 < KS && chain[s];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double jm = tile[k * kPlane + o_jm], jp = tile[k * kPlane + o_jp];
        const double im = tile[k * kPlane + o_im], ip = tile[k * kPlane + o_ip];
        const double c0 = f[s].own[k];
        const double km = k == 0 ? (up ? f[s > 0 ? s - 1 : 0].own[7] : f[s].km) : f[s].own[k > 0 ? k - 1 : 0];
        const double kp = k == 7 ? (dn ? f[s + 1 < KS ? s + 1 : s].own[0] : f[s].kp) : f[s].own[k < 7 ? k + 1 : 7];
        double res;
        if (MPI)
          res = (kp + km + jp + jm + ip + im) * cf.c[1] + c0 * cf.c[0];
        else
          res = cf.c[5] * kp + cf.c[6] * km + cf.c[3] * jp + cf.c[4] * jm + cf.c[1] * ip + cf.c[2] * im + cf.c[0] * c0;
        dst[k * 64 + lane] = res;
      }
    }
    if (s + 1 < KS) __syncthreads(); // all reads of level s are done before level s+1 overwrites the tiles
  }
}

// ---- variant 3: persistent waves, 3-deep pipeline ---------------------------------------------------------------
// Grid = 8 * wgs_per_xcd blocks.  XCD x (= blockIdx & 7, round-robin dispatch) owns list entries
// [x*chunk, (x+1)*chunk); its waves take entries lw, lw+W, lw+2W, ... (W = waves per XCD).  Entry indices past the
// end are clamped for the (harmless, redundant) prefetches and never computed.
template <bool MPI, bool HAS_IDS, int MINW>
__global__ __launch_bounds__(kWG, MINW) void k_s7_persist(const uint32_t *__restrict__ adj, const double *__restrict__ in,
                                                        double *__restrict__ out, size_t step,
                                                        const uint32_t *__restrict__ ids, uint32_t first, uint32_t n,
                                                        uint32_t chunk, Coeffs cf) {
  __shared__ double lds[kWavesPerWG * kTile];
  const int lane = threadIdx.x & 63;
  double *tile = lds + (threadIdx.x >> 6) * kTile;
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t W = (gridDim.x >> 3) * kWavesPerWG;
  // wave index made provably uniform: everything derived from it (entry index, trip count) stays scalar
  const uint32_t lw = (blockIdx.x >> 3) * kWavesPerWG + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t cb = xcd * chunk;
  const uint32_t ce = cb + chunk < n ? cb + chunk : n;
  uint32_t s = cb + lw;
  if (s >= ce) return;
  const uint32_t last = ce - 1;
  // brick id of a list entry as a (wave-uniform) vector value; made scalar only one trip later, so the wave never
  // waits on this load
  auto load_id = [&](uint32_t slot) -> uint32_t {
    const uint32_t c = slot < last ? slot : last;
    return HAS_IDS ? ids[c] : first + c;
  };
  // prologue: ids of entries e, e+1, e+2; adjacency rows of e, e+1; data of e
  uint32_t b_cur = __builtin_amdgcn_readfirstlane(load_id(s));
  uint32_t b_nxt = __builtin_amdgcn_readfirstlane(load_id(s + W));
  uint32_t vbA = load_id(s + 2 * W), vbB;
  Nbr nbA = load_nbr(adj, b_nxt), nbB;
  Faces A, B;
  issue_loads(A, in, step, b_cur, load_nbr(adj, b_cur), lane);

  // One pipeline phase: entry e is in CUR; prefetch the data of e+1 into NXT, the adjacency row of e+2 and the id
  // of e+3; then compute and store e.  Issue order: [id e+3][row e+2][14 data loads e+1][8 stores e], so the next
  // phase's first wait is vmcnt(8): it never waits for its predecessor's stores.
  auto phase = [&](Faces &CUR, Faces &NXT, const Nbr &nb_in, uint32_t vb_in, Nbr &nb_out, uint32_t &vb_out) {
    stage(tile, CUR, lane);
    vb_out = load_id(s + 3 * W);
    const uint32_t b_nn = __builtin_amdgcn_readfirstlane(vb_in);
    nb_out = load_nbr(adj, b_nn);
    issue_loads(NXT, in, step, b_nxt, nb_in, lane);
    __builtin_amdgcn_sched_barrier(0); // nothing moves across: the loads stay ahead of the compute block
    wave_lds_sync();
    compute_store<MPI>(tile, CUR, out + (size_t)b_cur * step, lane, cf);
    wave_lds_sync(); // every read of the tile is issued before the next stage overwrites it
    s += W;
    b_cur = b_nxt;
    b_nxt = b_nn;
  };
  // Shape of the loop, and why:
  //  * two named register sets, body unrolled by two -> no loaded value is ever copied (a copy would wait for it);
  //  * no branch inside the body -> hipcc cannot sink the prefetch below the compute block (it would, were the
  //    prefetch dead on an exit path), and waits are not merged conservatively over a skipped block;
  //  * the first phase is peeled so that both edges into the loop header carry the same pending-operation
  //    pattern (loads, then 8 stores); otherwise the header wait degrades to vmcnt(0), i.e. waits for stores.
  // Prefetches past the end of the chunk are clamped (redundant, harmless); an odd tail is handled after the loop.
  const uint32_t trips = (ce - s + W - 1) / W;
  phase(A, B, nbA, vbA, nbB, vbB);
  const uint32_t rest = trips - 1;
  for (uint32_t pair = 0; pair < (rest >> 1); ++pair) {
    phase(B, A, nbB, vbB, nbA, vbA);
    phase(A, B, nbA, vbA, nbB, vbB);
  }
  if (rest & 1) {
    stage(tile, B, lane);
    wave_lds_sync();
    compute_store<MPI>(tile, B, out + (size_t)__builtin_amdgcn_readfirstlane(b_cur) * step, lane, cf);
  }
}

} // namespace

int launch_star7(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step,
                 const uint32_t *ids, uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st) {
  const bool mpi = stencil_id == BRK_MPI7;
  if (variant == 3 || variant == 4) {
    // variant 3: <=96 VGPRs, 5 workgroups (20 waves) per CU; variant 4: <=128 VGPRs, 4 workgroups per CU
    const uint32_t per_cu = variant == 3 ? 5 : 4;
    uint32_t wgs = 256 * per_cu;
    const uint32_t need = (n + kWavesPerWG - 1) / kWavesPerWG;
    if (need < wgs) wgs = ((need + 7) / 8) * 8;
    const uint32_t chunk = (n + 7) / 8;
#define BRK_LAUNCH_PERSIST(M, I, W) \
  hipLaunchKernelGGL((k_s7_persist<M, I, W>), dim3(wgs), dim3(kWG), 0, st, adj, in, out, step, ids, first, n, chunk, cf)
#define BRK_LAUNCH_PERSIST_W(W)                      \
  do {                                               \
    if (mpi && ids) BRK_LAUNCH_PERSIST(true, true, W);    \
    else if (mpi) BRK_LAUNCH_PERSIST(true, false, W);     \
    else if (ids) BRK_LAUNCH_PERSIST(false, true, W);     \
    else BRK_LAUNCH_PERSIST(false, false, W);             \
  } while (0)
    if (variant == 3) BRK_LAUNCH_PERSIST_W(5);
    else BRK_LAUNCH_PERSIST_W(4);
#undef BRK_LAUNCH_PERSIST_W
#undef BRK_LAUNCH_PERSIST
  } else if (variant == 5 || variant == 6) {
    const uint32_t nw = variant == 5 ? 8 : 4;
    const uint32_t wgs = (((n + nw - 1) / nw + 7) / 8) * 8;
    if (variant == 5) {
      if (mpi) hipLaunchKernelGGL((k_s7_wg<true, 8>), dim3(wgs), dim3(512), 0, st, adj, in, out, step, ids, first, n, cf);
      else hipLaunchKernelGGL((k_s7_wg<false, 8>), dim3(wgs), dim3(512), 0, st, adj, in, out, step, ids, first, n, cf);
    } else {
      if (mpi) hipLaunchKernelGGL((k_s7_wg<true, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, cf);
      else hipLaunchKernelGGL((k_s7_wg<false, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, cf);
    }
  } else if (variant == 7 || variant == 8) {
    const uint32_t ks = variant == 7 ? 2 : 4;
    const uint32_t per_wg = 8 * ks;
    const uint32_t wgs = (((n + per_wg - 1) / per_wg + 7) / 8) * 8;
#define BRK_LAUNCH_WGK(M, K) hipLaunchKernelGGL((k_s7_wgk<M, K>), dim3(wgs), dim3(512), 0, st, adj, in, out, step, ids, first, n, cf)
    if (variant == 7) { if (mpi) BRK_LAUNCH_WGK(true, 2); else BRK_LAUNCH_WGK(false, 2); }
    else { if (mpi) BRK_LAUNCH_WGK(true, 4); else BRK_LAUNCH_WGK(false, 4); }
#undef BRK_LAUNCH_WGK
  } else if (variant >= 16 && variant < 24) {
    const uint32_t wgs = (((n + kWavesPerWG - 1) / kWavesPerWG + 7) / 8) * 8;
#define BRK_DIAG(S) case 16 + S: hipLaunchKernelGGL((k_s7<false, S>), dim3(wgs), dim3(kWG), 0, st, adj, in, out, step, ids, first, n, cf); break
    switch (variant) { BRK_DIAG(0); BRK_DIAG(1); BRK_DIAG(2); BRK_DIAG(3); BRK_DIAG(4); BRK_DIAG(5); BRK_DIAG(6); BRK_DIAG(7); }
#undef BRK_DIAG
  } else {
    const uint32_t wgs = (((n + kWavesPerWG - 1) / kWavesPerWG + 7) / 8) * 8;
    if (mpi)
      hipLaunchKernelGGL(k_s7<true>, dim3(wgs), dim3(kWG), 0, st, adj, in, out, step, ids, first, n, cf);
    else
      hipLaunchKernelGGL(k_s7<false>, dim3(wgs), dim3(kWG), 0, st, adj, in, out, step, ids, first, n, cf);
  }
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

} // namespace brk
