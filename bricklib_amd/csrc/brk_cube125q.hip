// brk_cube125q.hip -- the 125-point stencil at FOUR waves per SIMD (round 5): k_cube125_q.
//
// k_cube_dense<kCube125, 4, 3, 5> (brk_cubedense.hip) holds 166 VGPRs and 52 240 B of LDS per 4-wave group: 12 waves per CU.  Its three
// resources -- fp64 issue 0.26 ms, LDS pipe 0.29 ms, line requests 0.38 ms per 512^3 sweep -- overlap only across the waves of a CU, so a
// fourth wave per SIMD is worth 5 % (timing diagnostic profiles/r05_b_125pt_four_waves_per_simd_diagnostic.txt).  What it takes:
//   * <= 128 VGPRs.  The accumulation (cube125 output-side form, include/brick-device.h) needs ~86; what pushed the general kernel to 166
//     and made hipcc spill under a 128 cap was its fetch phase (break flags, two phases, conditional gathers on both sides of every wave).
//     Here the fetch phase is LEAN: groups of four fully linked i-neighbours only, so the two interior waves gather nothing on their
//     i-sides, and the adjacency row of a brick is arithmetic on its id.  The launcher selects this kernel only where that is PROVEN:
//     box entry point, registered adjacency table, rows that are multiples of 8 bricks, every brick of the box "regular" (all 26
//     neighbours are the ids the lexicographic numbering predicts: one reduction over the cube-regular mask per box and registration,
//     cached).  Everything else runs k_cube_dense as before.
//   * <= 40 960 B of LDS per group (160 KiB / 4), EXACTLY.  Four dense tiles D[12][12][8] are 36 864 B; the halo columns of the group's
//     two ends (2 columns x 144 rows x 2 sides = 4 608 B) and the output-side shift slots (6 144 B) do not fit beside them.  So:
//       - the group is laid out PLANE BY PLANE: plane p = [tile 0 | tile 1 | tile 2 | tile 3 | F], tile planes of 12 rows x 8 columns, F =
//         rows x 4 columns (i = -2, -1 of the left end, i = 8, 9 of the right end).  Tile cells and F cells of a plane share the plane's
//         offset, so one instruction with one immediate serves lanes that read a tile and lanes that read F;
//       - the F part of the four k-halo planes (0, 1, 10, 11) holds rows 2..9 only (32 doubles instead of 48): 8 x 432 + 4 x 416 = 5 120
//         doubles = 40 960 B.  The 64 cells that drops -- the eight 2 x 2 x 2 corner blocks of the halo on the group's two i-sides -- are
//         read only by the halo pass of the end waves' edge lanes; those lanes load them straight into registers (two 16-byte loads) and
//         select them in (four plane steps);
//       - the shift slots are columns 2..5 of planes 2..5 of the wave's OWN tile: read by no other wave (neighbours read columns 0, 1 /
//         6, 7), and dead for this wave once plane 5 has been requested, which is before output 0 is emitted.  Only 4 of a tile row's 8
//         columns are private, so the 16-lane write groups and 32-lane read groups of a slot access meet two-way in the banks (PMC:
//         SQ_LDS_BANK_CONFLICT 215 cycles per wave against 27 in k_cube_dense, LDS-array cycles +35 %): the price of 6 KiB of LDS.
//         Moving the partial sums between lanes instead (DPP row_ror:8 + ds_swizzle xor 16 + ds_bpermute, 10 crossbar operations per
//         output, no slot memory) was built and measured: the same time (0.5233-0.5334 vs 0.5275-0.5322 ms) -- profiles/EXPERIMENTS.md.
// Arithmetic and summation order are those of cube125_accumulate_jout (stencils/fake.h:87-341 folded over its 10 coefficient classes).
#include "brk_device.h"

namespace brklib {
namespace {

struct QL { // one 4-wave group in LDS, offsets in doubles
  static constexpr int NW = 4;
  __host__ __device__ static constexpr bool khalo(int p) { return p < 2 || p >= 10; }
  __host__ __device__ static constexpr int psize(int p) { return NW * 96 + (khalo(p) ? 32 : 48); }
  __host__ __device__ static constexpr int off(int p) {
    int o = 0;
    for (int q = 0; q < p; ++q) o += psize(q);
    return o;
  }
  static constexpr int TOTAL = 8 * (NW * 96 + 48) + 4 * (NW * 96 + 32);
  static constexpr int F = NW * 96; // F part of a plane, relative to the plane
  // F rows: the 8 rows beside the brick first (tile rows 2..9 -> 0..7), then the j-halo rows (tile rows 0, 1, 10, 11 -> 8..11; planes 2..9 only)
  __host__ __device__ static constexpr int frow(int row) { return row >= 2 && row < 10 ? row - 2 : (row < 2 ? 8 + row : row); }
  __host__ __device__ static constexpr int tile(int p, int row, int col) { return off(p) + row * 8 + col; }                   // + w * 96
  __host__ __device__ static constexpr int fcell(int side, int p, int row, int c) { return off(p) + F + frow(row) * 4 + side * 2 + c; }
};
static_assert(QL::off(12) == QL::TOTAL && QL::TOTAL * 8 == 40960, "the group fills a quarter of the CU's LDS exactly");
static_assert(QL::off(3) - QL::off(2) == 432 && QL::off(5) - QL::off(4) == 432, "shift slots: planes 2..5 have the pitch of the mid planes");

// Gather descriptors, one per lane and group: destination offset (doubles, pairs: even) | adjacency slot << 16 | source element << 21.
//   group 0: the kj-edge pairs of the own tile (k-halo plane x j-halo row, 4 pairs of columns): 64, relative to the wave's tile
//   group 1 / 2: the ki- and ji-edge pairs on the left / right i-side (F cells): 32 + 32 each; the corner blocks are NOT here
struct QTable {
  alignas(16) uint32_t e[64 * 4]; // lane-major: {own, left, right, unused}
};
constexpr QTable make_qtable() {
  QTable t{};
  for (int q = 0; q < 256; ++q) t.e[q] = 0xFFFFFFFFu;
  int n0 = 0, n1 = 0, n2 = 0;
  for (int kk = 0; kk < 12; ++kk)
    for (int jj = 0; jj < 12; ++jj) {
      const int dk = kk < 2 ? -1 : (kk >= 10 ? 1 : 0), dj = jj < 2 ? -1 : (jj >= 10 ? 1 : 0);
      const int sk = (kk - 2 + 8) % 8, sj = (jj - 2 + 8) % 8;
      if (dk != 0 && dj != 0) // own kj-edges: four pairs of columns
        for (int c = 0; c < 8; c += 2)
          t.e[4 * n0++ + 0] = (uint32_t)QL::tile(kk, jj, c) | ((uint32_t)adj_slot(dk, dj, 0) << 16) | ((uint32_t)(sk * 64 + sj * 8 + c) << 21);
      if ((dk != 0) != (dj != 0)) { // i-side edges (not faces, not corners): one pair per side
        t.e[4 * n1++ + 1] = (uint32_t)QL::fcell(0, kk, jj, 0) | ((uint32_t)adj_slot(dk, dj, -1) << 16) | ((uint32_t)(sk * 64 + sj * 8 + 6) << 21);
        t.e[4 * n2++ + 2] = (uint32_t)QL::fcell(1, kk, jj, 0) | ((uint32_t)adj_slot(dk, dj, 1) << 16) | ((uint32_t)(sk * 64 + sj * 8 + 0) << 21);
      }
    }
  // 64 lanes each (checked below through the counts the loops must reach)
  return (n0 == 64 && n1 == 64 && n2 == 64) ? t : QTable{};
}
__device__ const QTable g_qtable = make_qtable();
static_assert(make_qtable().e[0] != 0 || make_qtable().e[1] != 0, "descriptor table: 64 pairs per group");

typedef double d2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) d2 *gc2;

// One pass of the output-side accumulation over NP planes (see cube125_jout_stream, include/brick-device.h): rd(p, d) is the value in
// column i + d - 2 of the lane's row in plane p of the pass; fix(p, x) patches the five values of a plane (corner cells).
template <int NP, bool MAIN, class Rd, class Fix, class Emit>
__device__ __forceinline__ void q_stream(const Coeffs &cf, Rd rd, Fix fix, Emit emit) {
  constexpr int wa[6] = {0, 1, 2, 0, 0, 1}, wb[6] = {0, 1, 2, 1, 2, 2};
  double x[2][5]; // rows in flight
  double V[5][3]; // |di|-folded rows of the last five planes
#pragma unroll
  for (int d = 0; d < 5; ++d) x[0][d] = rd(0, d);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p + 1 < NP) {
#pragma unroll
      for (int d = 0; d < 5; ++d) x[(p + 1) & 1][d] = rd(p + 1, d);
    }
    __builtin_amdgcn_sched_barrier(0);
    fix(p, x[p & 1]);
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

// PRED: the adjacency row is computed from the brick id and the strides of the box numbering (every brick this kernel sees is regular)
template <int PRED>
__global__ __launch_bounds__(256, 4) void k_cube125_q(const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out,
                                                      size_t step, const uint32_t *__restrict__ ids, uint32_t n, Coeffs cf, uint32_t cs1, uint32_t cs2) {
  __shared__ double lds[QL::TOTAL];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wg * QL::NW + w;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  const int l5 = lane & 31, lp = lane >> 5;
  const uint32_t b = slot < n ? __builtin_amdgcn_readfirstlane(ids[slot]) : BRK_LIST_SKIP;
  if (b == BRK_LIST_SKIP) return; // padding comes in whole groups (rows are multiples of 8 entries): all four waves leave together
  const bool endL = w == 0, endR = w == QL::NW - 1;
  double *const tile = lds + w * 96; // + QL::tile(p, row, col)

  // ---- fetch: everything is requested before anything is staged
  const int l27 = lane < 27 ? lane : 13;
  uint32_t arow;
  if (PRED) arow = b + (uint32_t)(l27 / 9 - 1) * cs2 + (uint32_t)((l27 / 3) % 3 - 1) * cs1 + (uint32_t)(l27 % 3 - 1);
  else arow = adj[(size_t)b * 27 + l27];
  const gcptr self = sbase(in + (size_t)b * step);
  d2 own2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) own2[q] = *(gc2)(self + q * 128 + 2 * lane); // lane l of instruction q: elements 2 (l % 32), +1 of plane 2q + l / 32
  if (!PRED) __builtin_amdgcn_sched_barrier(0);
  const gcptr p_km = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)) * step);
  const gcptr p_kp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0)) * step);
  const gcptr p_jm = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0)) * step);
  const gcptr p_jp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0)) * step);
  const d2 km2 = *(gc2)(p_km + 6 * 64 + 2 * lane), kp2 = *(gc2)(p_kp + 2 * lane);                   // planes 6, 7 below / 0, 1 above
  const d2 jm2 = *(gc2)(p_jm + hi * 64 + 48 + 2 * lo), jp2 = *(gc2)(p_jp + hi * 64 + 2 * lo);       // lane (k, chunk): rows 6, 7 / 0, 1 of plane k
  const uint4 dsc = reinterpret_cast<const uint4 *>(g_qtable.e)[lane];
  auto pair_at = [&](uint32_t d) -> d2 {
    const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((d >> 16) & 31u) << 2), (int)arow);
    return *(const d2 *)(in + (size_t)nb * step + ((d >> 21) & 511u));
  };
  const d2 g0 = pair_at(dsc.x);
  // the group's two end waves: the i-face columns, the i-side edge pairs, and the corner cells of the halo pass (registers)
  const int kh = lane >> 5, jh = (lane >> 3) & 3;
  d2 side = {0.0, 0.0}, gs = {0.0, 0.0}, cn[2] = {{0.0, 0.0}, {0.0, 0.0}};
  if (endL || endR) { // wave-uniform; the left end takes i = -2, -1 (columns 6, 7 of the bricks on that side), the right end i = 8, 9 (columns 0, 1)
    const int di = endL ? -1 : 1, c0 = endL ? 6 : 0;
    side = *(gc2)(sbase(in + (size_t)__builtin_amdgcn_readlane(arow, endL ? adj_slot(0, 0, -1) : adj_slot(0, 0, 1)) * step) + hi * 64 + lo * 8 + c0);
    gs = pair_at(endL ? dsc.y : dsc.z);
    // corner cells of lane (kh, jh, .): planes kh ? 10, 11 : 0, 1, row hrow(jh): brick (kh ? +1 : -1, jh < 2 ? -1 : +1, di)
    const int cslot = adj_slot(kh ? 1 : -1, jh < 2 ? -1 : 1, di);
    const uint32_t cb = (uint32_t)__builtin_amdgcn_ds_bpermute(cslot << 2, (int)arow);
    const double *src = in + (size_t)cb * step + (kh ? 0 : 6) * 64 + (jh < 2 ? 6 + jh : jh - 2) * 8 + c0;
    cn[0] = *(const d2 *)src, cn[1] = *(const d2 *)(src + 64);
  }

  // ---- stage
#pragma unroll
  for (int q = 0; q < 4; ++q) *(d2 *)&tile[QL::off(2 * q + 2) + lp * 432 + ((l5 >> 2) + 2) * 8 + 2 * (l5 & 3)] = own2[q]; // planes 2..9: pitch 432
  *(d2 *)&tile[lp * 416 + ((l5 >> 2) + 2) * 8 + 2 * (l5 & 3)] = km2;                                                      // planes 0, 1
  *(d2 *)&tile[QL::off(10) + lp * 416 + ((l5 >> 2) + 2) * 8 + 2 * (l5 & 3)] = kp2;                                        // planes 10, 11
  *(d2 *)&tile[QL::off(2) + hi * 432 + 2 * lo] = jm2;                                                                     // rows 0, 1 of plane hi + 2
  *(d2 *)&tile[QL::off(2) + hi * 432 + 10 * 8 + 2 * lo] = jp2;                                                            // rows 10, 11
  *(d2 *)&tile[dsc.x & 0xFFFFu] = g0;
  if (endL || endR) {
    const int s = endL ? 0 : 1;
    *(d2 *)&lds[QL::off(2) + hi * 432 + QL::F + lo * 4 + s * 2] = side; // lane (k, j): F row j of plane k + 2
    *(d2 *)&lds[(endL ? dsc.y : dsc.z) & 0xFFFFu] = gs;
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);

  // ---- accumulate (output-side form): column i + d - 2 of a row is a tile cell of this wave, of a sibling wave, or an F cell
  gptr dst = (gptr)(out + (size_t)b * step);
  asm volatile("" : "+s"(dst));
  // main rows: row hi + 2 of every plane; offsets relative to lds + plane offset
  int com[5], coh[5];
  const int hrow = jh < 2 ? jh : jh + 8; // halo rows 0, 1, 10, 11
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const int c = lo + d - 2;
    const bool fl = c < 0 && endL, fr = c > 7 && endR;
    const int tcol = (int)w * 96 + c; // a sibling's tile is 96 doubles away: w * 96 + c runs through the neighbours' columns by itself
    com[d] = fl ? QL::F + hi * 4 + (c + 2) : (fr ? QL::F + hi * 4 + 2 + (c - 8) : (c < 0 ? tcol - 96 + 8 : (c > 7 ? tcol + 96 - 8 : tcol)) + (hi + 2) * 8);
    coh[d] = fl ? QL::F + QL::frow(0) * 4 + (hrow < 2 ? hrow : hrow - 8) * 4 + (c + 2)
                : (fr ? QL::F + QL::frow(0) * 4 + (hrow < 2 ? hrow : hrow - 8) * 4 + 2 + (c - 8) : (c < 0 ? tcol - 96 + 8 : (c > 7 ? tcol + 96 - 8 : tcol)) + hrow * 8);
  }
  // halo pass: lanes with kh = 1 stream planes 4..11, i.e. plane t + 4 in step t; off(t + 4) - off(t) takes three values
  int cob[3][5];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    cob[0][d] = coh[d] + kh * (QL::off(4) - QL::off(0));
    cob[1][d] = coh[d] + kh * (QL::off(5) - QL::off(1));
    cob[2][d] = coh[d] + kh * (QL::off(6) - QL::off(2));
  }
  static_assert(QL::off(11) - QL::off(7) == QL::off(5) - QL::off(1) && QL::off(7) - QL::off(3) == QL::off(6) - QL::off(2) &&
                    QL::off(10) - QL::off(6) == QL::off(6) - QL::off(2) && QL::off(9) - QL::off(5) == QL::off(6) - QL::off(2) &&
                    QL::off(8) - QL::off(4) == QL::off(6) - QL::off(2), "three plane-offset classes");
  double h1[4], h2[4];
  // corner patch: in the steps whose plane is a k-halo plane for this lane's half (kh = 0: steps 0, 1; kh = 1: steps 6, 7) the i-halo
  // columns of the j-halo rows are not in LDS: lanes lo < 2 of the left end / lo > 5 of the right end take them from cn[]
  const bool cl0 = endL && lo == 0, cl1 = endL && lo == 1, cr6 = endR && lo == 6, cr7 = endR && lo == 7;
  q_stream<8, false>(
      cf,
      [&](int t, int d) __attribute__((always_inline)) { return lds[cob[t == 0 ? 0 : (t == 1 || t == 7 ? 1 : 2)][d] + QL::off(t)]; },
      [&](int t, double(&x)[5]) __attribute__((always_inline)) {
        if (!(t < 2 || t >= 6)) return;
        if (!(endL || endR)) return; // wave-uniform
        const bool mine = (t < 2) == (kh == 0);
        const d2 c = cn[t & 1]; // steps 0, 1 -> planes 0, 1 (kh = 0); steps 6, 7 -> planes 10, 11 (kh = 1)
        if (mine && cl0) x[0] = c.x, x[1] = c.y; // columns -2, -1
        if (mine && cl1) x[0] = c.y;             // column -1
        if (mine && cr6) x[4] = c.x;             // column 8
        if (mine && cr7) x[3] = c.x, x[4] = c.y; // columns 8, 9
      },
      [&](int k, double, double b1, double b2) __attribute__((always_inline)) {
        asm volatile("" : "+v"(b1), "+v"(b2)); // computed here, by all lanes, not sunk into the half-wave branch that uses it
        h1[k] = b1, h2[k] = b2;
      });
  double pb0 = 0.0, r[4] = {0.0, 0.0, 0.0, 0.0};
  auto retire = [&](int k) __attribute__((always_inline)) { store_result(&dst[k * 64 + lane], pb0 + ((r[0] + r[1]) + (r[2] + r[3]))); };
  // shift slots: slot (row r, column c) = plane (4 | 2) + c / 4 of the own tile, row r, column 2 + c % 4
  double *const s1 = tile + QL::off(4) + (lo >> 2) * 432 + 2 + (lo & 3), *const s2 = tile + QL::off(2) + (lo >> 2) * 432 + 2 + (lo & 3);
  q_stream<12, true>(
      cf, [&](int p, int d) __attribute__((always_inline)) { return lds[com[d] + QL::off(p)]; }, [&](int, double(&)[5]) __attribute__((always_inline)) {},
      [&](int k, double b0, double b1, double b2) __attribute__((always_inline)) {
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

// are all bricks of a list regular (bit set in the cube-regular mask of the registered table)?  *flag &= all
__global__ void k_all_regular(const uint32_t *__restrict__ ids, uint32_t n, const uint32_t *__restrict__ mask, uint32_t *flag) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = true;
  if (q < n) {
    const uint32_t b = ids[q];
    ok = b == BRK_LIST_SKIP || ((mask[b >> 5] >> (b & 31u)) & 1u);
  }
  if (__ballot(!ok) != 0ull && (threadIdx.x & 63) == 0) atomicAnd(flag, 0u);
}

} // namespace

int all_regular(const uint32_t *ids, uint32_t n, const uint32_t *mask, hipStream_t st, bool *out) {
  uint32_t *flag = nullptr, host = 1u;
  HIP_TRY(hipMalloc(&flag, sizeof(uint32_t)));
  hipError_t r = hipMemcpyAsync(flag, &host, sizeof host, hipMemcpyHostToDevice, st);
  if (r == hipSuccess) {
    hipLaunchKernelGGL(k_all_regular, dim3((n + 255) / 256), dim3(256), 0, st, ids, n, mask, flag);
    r = hipGetLastError();
  }
  if (r == hipSuccess) r = hipMemcpyAsync(&host, flag, sizeof host, hipMemcpyDeviceToHost, st);
  if (r == hipSuccess) r = hipStreamSynchronize(st);
  (void)hipFree(flag);
  HIP_TRY(r);
  *out = host != 0u;
  return BRK_OK;
}

int launch_cube125_q(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, uint32_t n, const Coeffs &cf, hipStream_t st,
                     uint32_t s1, uint32_t s2, bool predicted) {
  const uint32_t wgs4 = (((n + 3) / 4 + 7) / 8) * 8;
  brki_note_kernel(predicted ? "k_cube125_q<1>" : "k_cube125_q<0>");
  if (predicted) hipLaunchKernelGGL((k_cube125_q<1>), dim3(wgs4), dim3(256), 0, st, adj, in, out, step, ids, n, cf, s1, s2);
  else hipLaunchKernelGGL((k_cube125_q<0>), dim3(wgs4), dim3(256), 0, st, adj, in, out, step, ids, n, cf, s1, s2);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

} // namespace brklib
