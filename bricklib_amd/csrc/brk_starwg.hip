// brk_starwg.hip -- star stencils of radius 2 and 4 (13- and 25-point) with workgroup-shared i-halos.
//
// Same idea as k_s7_wg (brk_star7.hip): a 512-thread workgroup takes 8 consecutive list entries, each wave stages its
// brick (+ the R rows of the two j-neighbours) in a dense LDS tile, and the R columns it needs from its i-neighbours
// are read straight out of the sibling waves' tiles.  That removes the 2R gather instructions per brick (64 lanes x
// 8 B at 64-B stride, every line of the neighbour brick) that dominated the one-wave-per-brick version.  A wave whose
// list neighbour is not its i-neighbour (row ends, irregular lists) gathers the columns into its own fallback slots.
// k-neighbours stay in registers (a column of 8+2R values per lane).
#include "brk_device.h"
#include "brk_starwg_kernel.h"

namespace brklib {
namespace {

#ifdef BRK_VARIANTS
#define BRK_VARIANTS_PART 1
#include "brk_starwg_variants.hip"
#undef BRK_VARIANTS_PART
#endif

// ---- NB k-consecutive bricks per wave: the k-halo is carried, not fetched ---------------------------------------------------------
// k_star_kp shares the planes BETWEEN the two bricks of a pair; a radius-4 wave still requests 24 planes for 16 of its own.  Here a
// workgroup of 4 waves takes 4 NB list entries -- NB runs of four i-neighbours, run n+1 the +k neighbours of run n (lists in the
// k-stacked order: bricklib_amd/order.py, the box entry point) -- and wave w marches up its column of NB bricks with ONE register
// column: the R planes below brick 0 and above brick NB-1 are the only halo planes it ever requests (8 NB + 2 R planes per column
// instead of NB (8 + 2 R)), a brick's upper halo being the first planes of the next brick, which are in flight anyway.  The loads of
// brick n+2 (and the faces of brick n+1) are issued BEFORE brick n is staged, computed and stored: one whole step of cover, and the
// wait for them does not wait for the stores issued after them (loads and stores share the in-order vmcnt counter; the fast path has
// no branch around its loads and stores, so the compiler counts exactly).  One tile per wave, reused from brick to brick.
// The fast path needs every one of the 4 NB entries active and every column a k-chain; the workgroup votes once (uniform), and any
// other group -- list ends, ragged rows, lists in another order -- takes the slow path: NB passes of the one-brick body with private
// halos.  Results never depend on the path.
template <int KIND, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void k_star_km(
    const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out, size_t step,
    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf, const uint32_t *__restrict__ faces) {
  constexpr int R = StarTraits<KIND>::R, H = R / 2, NW = 4;
  using TL = WgTile<R>;
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) d2 *gc2;
  __shared__ double lds[NW * TL::TILE];
  __shared__ uint32_t vote[NW];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * TL::TILE;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };
  auto entry = [&](uint32_t sl) -> uint32_t { return sl < n ? __builtin_amdgcn_readfirstlane(id_of(sl)) : BRK_LIST_SKIP; };
  const uint32_t slot0 = wg * (NW * NB) + w; // entry (w, n) = slot0 + n * NW
  struct Nb6 {
    uint32_t km, kp, jm, jp, im, ip;
  };
  auto neighbours = [&](uint32_t b) {
    Nb6 r;
    if (faces) { // two branches, each through one restrict pointer: scalar loads (see k_star_wg)
      const uint32_t *f = faces + (size_t)b * 8;
      r.km = __builtin_amdgcn_readfirstlane(f[0]), r.kp = __builtin_amdgcn_readfirstlane(f[1]);
      r.jm = __builtin_amdgcn_readfirstlane(f[2]), r.jp = __builtin_amdgcn_readfirstlane(f[3]);
      r.im = __builtin_amdgcn_readfirstlane(f[4]), r.ip = __builtin_amdgcn_readfirstlane(f[5]);
    } else {
      const uint32_t *row = adj + (size_t)b * 27;
      r.km = __builtin_amdgcn_readfirstlane(row[adj_slot(-1, 0, 0)]), r.kp = __builtin_amdgcn_readfirstlane(row[adj_slot(1, 0, 0)]);
      r.jm = __builtin_amdgcn_readfirstlane(row[adj_slot(0, -1, 0)]), r.jp = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 1, 0)]);
      r.im = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 0, -1)]), r.ip = __builtin_amdgcn_readfirstlane(row[adj_slot(0, 0, 1)]);
    }
    return r;
  };
  uint32_t b[NB];
  Nb6 nb[NB];
  bool chain = true;
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    b[q] = entry(slot0 + q * NW);
    chain = chain && b[q] != BRK_LIST_SKIP;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) nb[q] = neighbours(chain ? b[q] : 0u); // brick 0 always exists (the dummy brick)
#pragma unroll
  for (int q = 0; q + 1 < NB; ++q) chain = chain && nb[q].kp == b[q + 1] && nb[q + 1].km == b[q];
  if (lane == 0) vote[w] = chain ? 1u : 0u;
  __syncthreads();
  const bool fast = (vote[0] & vote[1] & vote[2] & vote[3]) != 0u; // uniform over the workgroup

  const int ik = R == 2 ? hi : (lane >> 4), ij = R == 2 ? lo : ((lane >> 1) & 7), ih = R == 2 ? 0 : (lane & 1);
  struct FaceRegs {
    d2 jm[H], jp[H], im[H], ip[H];
  };
  // share flags of brick q: is the list neighbour in its run the i-neighbour?
  auto shares = [&](int q, bool &s_im, bool &s_ip) {
    const uint32_t prev = w > 0 ? entry(slot0 + q * NW - 1) : BRK_LIST_SKIP, next = w + 1 < NW ? entry(slot0 + q * NW + 1) : BRK_LIST_SKIP;
    s_im = prev != BRK_LIST_SKIP && nb[q].im == prev;
    s_ip = next != BRK_LIST_SKIP && nb[q].ip == next;
  };
  auto load_faces = [&](const Nb6 &r, bool s_im, bool s_ip, FaceRegs &f) {
    const gcptr p_jm = sbase(in + (size_t)r.jm * step), p_jp = sbase(in + (size_t)r.jp * step);
#pragma unroll
    for (int q = 0; q < H; ++q) {
      f.jm[q] = *(gc2)(p_jm + hi * 64 + (8 - R) * 8 + q * 16 + 2 * lo);
      f.jp[q] = *(gc2)(p_jp + hi * 64 + q * 16 + 2 * lo);
      f.im[q] = f.ip[q] = d2{0.0, 0.0};
    }
    if (!s_im) {
      const gcptr p_im = sbase(in + (size_t)r.im * step);
#pragma unroll
      for (int q = 0; q < H; ++q) f.im[q] = *(gc2)(p_im + (ik + 4 * q) * 64 + ij * 8 + (8 - R) + 2 * ih);
    }
    if (!s_ip) {
      const gcptr p_ip = sbase(in + (size_t)r.ip * step);
#pragma unroll
      for (int q = 0; q < H; ++q) f.ip[q] = *(gc2)(p_ip + (ik + 4 * q) * 64 + ij * 8 + 2 * ih);
    }
  };
  auto stage_faces = [&](const FaceRegs &f, bool s_im, bool s_ip) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      *(d2 *)&tile[hi * TL::PLANE + (2 * q) * 8 + 2 * lo] = f.jm[q];
      *(d2 *)&tile[hi * TL::PLANE + (R + 8 + 2 * q) * 8 + 2 * lo] = f.jp[q];
    }
    if (!s_im) {
#pragma unroll
      for (int q = 0; q < H; ++q) {
        tile[(ik + 4 * q) * TL::PLANE + TL::IM + (2 * ih) * 8 + ij] = f.im[q].x;
        tile[(ik + 4 * q) * TL::PLANE + TL::IM + (2 * ih + 1) * 8 + ij] = f.im[q].y;
      }
    }
    if (!s_ip) {
#pragma unroll
      for (int q = 0; q < H; ++q) {
        tile[(ik + 4 * q) * TL::PLANE + TL::IP + (2 * ih) * 8 + ij] = f.ip[q].x;
        tile[(ik + 4 * q) * TL::PLANE + TL::IP + (2 * ih + 1) * 8 + ij] = f.ip[q].y;
      }
    }
  };
  // out planes of brick `bb` from the tile and the lane's k-column colv(idx), idx in [0, 8 + 2R), own planes at [R, R + 8)
  auto compute = [&](uint32_t bb, bool s_im, bool s_ip, auto colv) {
    gptr dst = (gptr)(out + (size_t)bb * step);
    asm volatile("" : "+s"(dst));
    int o_im[R], o_ip[R];
#pragma unroll
    for (int q = 1; q <= R; ++q) {
      const int cm = lo - q, cp = lo + q;
      o_im[q - 1] = cm >= 0 ? (hi + R) * 8 + cm : (s_im ? -TL::TILE + (hi + R) * 8 + cm + 8 : TL::IM + (cm + R) * 8 + hi);
      o_ip[q - 1] = cp < 8 ? (hi + R) * 8 + cp : (s_ip ? TL::TILE + (hi + R) * 8 + cp - 8 : TL::IP + (cp - 8) * 8 + hi);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double jm[R], jp[R], im[R], ip[R];
#pragma unroll
      for (int q = 1; q <= R; ++q) {
        jm[q - 1] = tile[k * TL::PLANE + (hi + R - q) * 8 + lo];
        jp[q - 1] = tile[k * TL::PLANE + (hi + R + q) * 8 + lo];
        im[q - 1] = tile[k * TL::PLANE + o_im[q - 1]];
        ip[q - 1] = tile[k * TL::PLANE + o_ip[q - 1]];
      }
      double res = 0.0; // stencils/fake.h:41-53 (25-pt) / :73-79 (13-pt): outermost shell first, centre last
#pragma unroll
      for (int q = R; q >= 1; --q) {
        const double sh = colv(R + k + q) + colv(R + k - q) + jp[q - 1] + jm[q - 1] + ip[q - 1] + im[q - 1];
        res = (q == R) ? sh * cf.c[q] : res + sh * cf.c[q];
      }
      res = res + colv(R + k) * cf.c[0];
      store_result(&dst[k * 64 + lane], res);
    }
  };

  if (fast) {
    double P[8 * NB + 2 * R]; // the column: [0, R) below brick 0 | 8 NB own planes | [R + 8 NB, +R) above brick NB - 1
    FaceRegs F[2];
    bool sm[NB], sp[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) shares(q, sm[q], sp[q]);
    auto load_own = [&](int q) {
      const gcptr self = sbase(in + (size_t)b[q] * step);
#pragma unroll
      for (int k = 0; k < 8; ++k) P[R + 8 * q + k] = self[k * 64 + lane];
    };
    {
      const gcptr p = sbase(in + (size_t)nb[0].km * step);
#pragma unroll
      for (int r = 0; r < R; ++r) P[r] = p[(8 - R + r) * 64 + lane];
    }
    load_own(0);
    load_faces(nb[0], sm[0], sp[0], F[0]);
    if (NB > 1) load_own(1);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      // requests of the next step first: own planes of brick q + 2 (or the planes above the column), faces of brick q + 1
      if (q + 2 < NB) load_own(q + 2);
      if (q + 2 == NB || NB == 1) {
        const gcptr p = sbase(in + (size_t)nb[NB - 1].kp * step);
#pragma unroll
        for (int r = 0; r < R; ++r) P[R + 8 * NB + r] = p[r * 64 + lane];
      }
      if (q + 1 < NB) load_faces(nb[q + 1], sm[q + 1], sp[q + 1], F[(q + 1) & 1]);
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[k * TL::PLANE + (hi + R) * 8 + lo] = P[R + 8 * q + k];
      stage_faces(F[q & 1], sm[q], sp[q]);
      __syncthreads(); // the tiles of step q are complete
      compute(b[q], sm[q], sp[q], [&](int idx) __attribute__((always_inline)) { return P[8 * q + idx]; });
      if (q + 1 < NB) __syncthreads(); // every wave has finished reading the tiles of step q
    }
    return;
  }
  // ---- slow path: one brick at a time, private halos (any list)
#pragma unroll 1
  for (int q = 0; q < NB; ++q) {
    const uint32_t slot = slot0 + q * NW;
    const uint32_t bb = entry(slot);
    const bool active = bb != BRK_LIST_SKIP;
    double col[8 + 2 * R];
    bool s_im = false, s_ip = false;
    if (active) {
      const Nb6 r = neighbours(bb);
      const uint32_t prev = w > 0 ? entry(slot - 1) : BRK_LIST_SKIP, next = w + 1 < NW ? entry(slot + 1) : BRK_LIST_SKIP;
      s_im = prev != BRK_LIST_SKIP && r.im == prev;
      s_ip = next != BRK_LIST_SKIP && r.ip == next;
      const gcptr self = sbase(in + (size_t)bb * step), p_km = sbase(in + (size_t)r.km * step), p_kp = sbase(in + (size_t)r.kp * step);
#pragma unroll
      for (int k = 0; k < 8; ++k) col[R + k] = self[k * 64 + lane];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        col[t] = p_km[(8 - R + t) * 64 + lane];
        col[R + 8 + t] = p_kp[t * 64 + lane];
      }
      FaceRegs f;
      load_faces(r, s_im, s_ip, f);
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[k * TL::PLANE + (hi + R) * 8 + lo] = col[R + k];
      stage_faces(f, s_im, s_ip);
    }
    __syncthreads();
    if (active) compute(bb, s_im, s_ip, [&](int idx) __attribute__((always_inline)) { return col[idx]; });
    __syncthreads();
  }
}

} // namespace

int launch_star_wg(int stencil_id, int variant, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st, const uint32_t *faces, bool kpaired) {
  // Smaller workgroups = more of them resident and less lockstep between the load and the compute phase; larger ones share more
  // i-faces.  Measured at 512^3: 13-pt 4 waves (32 KiB LDS, 5 groups = 20 waves/CU) 0.392 ms vs 8 waves (64 KiB, 16 waves/CU)
  // 0.408 ms; 25-pt (12 KiB of LDS per wave, 12 waves/CU either way) 4 waves 0.502 ms vs 2 waves 0.507 ms -- since every line of a
  // face is requested once the 25-pt i-faces are the dearer part (before that: 2 waves 0.532 vs 4 waves 0.545).  Variant 2 = the
  // other group size.
  // defaults: 13-point 4-wave groups (5 per CU), 25-point 4-wave groups; -DBRK_VARIANTS builds: variant 2 = the other group size
#ifdef BRK_VARIANTS
#define BRK_VARIANTS_PART 2
#include "brk_starwg_variants.hip"
#undef BRK_VARIANTS_PART
#endif
#ifdef BRK_DIAG // diagnostic build only (make EXTRA_HIPFLAGS=-DBRK_DIAG): these variants compute WRONG results on purpose
  if (stencil_id == BRK_MPI25 && variant >= 7 && variant <= 12) {
    // timing experiments: 7 = no k-face loads, 8 = no j-face loads, 9 = neither; 10 = k-faces read from the brick itself, 11 = j-faces, 12 = both
    const uint32_t wgs = (((n + 1) / 2 + 7) / 8) * 8;
#define BRK_DIAG25(V, D) if (variant == V) hipLaunchKernelGGL((k_star_wg<kStarMpi25, 2, 3, D>), dim3(wgs), dim3(128), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces)
    BRK_DIAG25(7, 1); BRK_DIAG25(8, 2); BRK_DIAG25(9, 3); BRK_DIAG25(10, 4); BRK_DIAG25(11, 8); BRK_DIAG25(12, 12);
#undef BRK_DIAG25
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
#endif
  // Two k-consecutive bricks per wave over lists the caller declares k-paired (the box entry point builds its own).
  // Round 3: k_star_kp (everything requested up front, one register column for the pair): 25-point 0.472 ms against 0.507 for
  // k_star_wg on the tiled list (profiles/r03_f_kpair_star_kernels.txt).  Round 4: k_star_km<NB = 2> (a march: brick 1's faces and
  // the planes above the pair are requested while brick 0 is staged and computed, no selects in the accumulation), interleaved on one
  // box (profiles/r04_c_star_kmarch_kernels.txt): 25-point 0.444-0.448 against 0.470 for k_star_kp (-5 %), 13-point 0.3845 against
  // 0.3918 for k_star_wg on its tiled list (-1.9 %; k_star_kp: 0.3924).  NB = 4 is slower (0.534 / 0.390: longer barrier-coupled
  // chains, 168 VGPRs with spills for the radius-4 star).  Lists that are NOT paired go through the march kernel's slow path (one
  // brick at a time: 0.565 / 0.445 ms) -- declare only what is true.  -DBRK_VARIANTS builds: variant 3 = k_star_kp.
  if (kpaired) {
    const uint32_t wgs8 = (((n + 7) / 8 + 7) / 8) * 8;
    brki_note_kernel(stencil_id == BRK_MPI13 ? "k_star_km<kStarMpi13,2,4>" : "k_star_km<kStarMpi25,2,3>");
    if (stencil_id == BRK_MPI13) hipLaunchKernelGGL((k_star_km<kStarMpi13, 2, 4>), dim3(wgs8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    else hipLaunchKernelGGL((k_star_km<kStarMpi25, 2, 3>), dim3(wgs8), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
  brki_note_kernel(stencil_id == BRK_MPI13 ? "k_star_wg<kStarMpi13,4,5>" : "k_star_wg<kStarMpi25,4,3>");
  if (stencil_id == BRK_MPI13)
    hipLaunchKernelGGL((k_star_wg<kStarMpi13, 4, 5>), dim3(wgs), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
  else
    hipLaunchKernelGGL((k_star_wg<kStarMpi25, 4, 3>), dim3(wgs), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, faces);
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

} // namespace brklib
