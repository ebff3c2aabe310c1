// brk_cubedense_variants.hip -- earlier kernel generations and experiments of brk_cubedense.hip: compiled ONLY in `make VARIANTS=1` builds (A/B runs
// with tools/kbench through brk_set_variant, and the variant parity tests); the default libbrk.so does not contain a line of this file.
// It is not a translation unit of its own: brk_cubedense.hip includes it twice under -DBRK_VARIANTS, once for the kernels (part 1, inside
// the anonymous namespace, so that they share the helpers of the shipped kernel) and once for the launch dispatch (part 2).
#if BRK_VARIANTS_PART == 1

// ---- LDS-DMA staging (round 4, k_cube_dma; MEASURED SLOWER, development builds only: profiles/r04_j_lds_dma_staging.txt): a tile is ELEMS * 8 contiguous bytes, so ceil(ELEMS / 128) instructions of 1 KiB
// (global_load_lds_dwordx4: 16 bytes per lane, the LDS side contiguous, the global side a per-lane address) fill ALL of it -- own
// rows, k- and j-faces and the kj-edge rows -- without staging registers and without ds_write.  One descriptor per (instruction,
// lane): where the 16 bytes that belong at that LDS position come from: adjacency slot << 16 | element offset inside that brick << 21.
// Rows the diamond never reads are pointed at the own brick (a line the wave requests anyway).
template <int KIND>
struct DmaX {
  using TL = DenseTile<KIND>;
  static constexpr int R = CubeTraits<KIND>::R;
  static constexpr int T = 8 + 2 * R;
  static constexpr int NI = (TL::ELEMS + 127) / 128;
  static constexpr int QUADS = (NI + 3) / 4;
};
template <int KIND>
struct DmaTable {
  alignas(16) uint32_t e[DmaX<KIND>::QUADS * 256];
};
template <int KIND>
constexpr DmaTable<KIND> make_dma_table() {
  DmaTable<KIND> t{};
  using X = DmaX<KIND>;
  using TL = DenseTile<KIND>;
  constexpr int R = X::R, T = X::T;
  for (int q = 0; q < X::QUADS * 256; ++q) t.e[q] = 0xFFFFFFFFu;
  for (int n = 0; n < X::NI * 64; ++n) {
    const int q = n / 64, lane = n % 64;
    const int e = q * 128 + lane * 2; // first of the two doubles this lane brings
    if (e >= TL::ELEMS) continue;
    const int row = e / 8, c = e % 8, kk = row / T + TL::KK0, jj = row % T;
    int dk = kk < R ? -1 : (kk >= R + 8 ? 1 : 0), dj = jj < R ? -1 : (jj >= R + 8 ? 1 : 0);
    if (KIND == kDiamond25) {
      const int ek = kk < R ? R - kk : (kk >= R + 8 ? kk - R - 7 : 0), ej = jj < R ? R - jj : (jj >= R + 8 ? jj - R - 7 : 0);
      if (ek + ej > 2) dk = 0, dj = 0; // never read: any address the wave touches anyway
    }
    const int sk = (kk - R + 8) % 8, sj = (jj - R + 8) % 8;
    t.e[((q / 4) * 64 + lane) * 4 + q % 4] = ((uint32_t)adj_slot(dk, dj, 0) << 16) | ((uint32_t)(sk * 64 + sj * 8 + c) << 21);
  }
  return t;
}
__device__ const DmaTable<kCube27> g_dma27 = make_dma_table<kCube27>();
__device__ const DmaTable<kCube125> g_dma125 = make_dma_table<kCube125>();
__device__ const DmaTable<kDiamond25> g_dma_d25 = make_dma_table<kDiamond25>();
template <int KIND>
__device__ __forceinline__ const uint32_t *dma_table() {
  if (KIND == kCube27) return g_dma27.e;
  if (KIND == kCube125) return g_dma125.e;
  return g_dma_d25.e;
}

// k_cube_dma (round 4) -- the same tiles and the same accumulation as k_cube_dense, for REGISTERED adjacency tables, staged by LDS-DMA.
// Correct (the whole GPU suite ran through it) and 10-19 % SLOWER than staging through registers: 27-point 0.465 vs 0.423 ms, 125-point
// 0.672 vs 0.564, fused double 7-point pass 0.506 vs 0.437 (interleaved, placed storages, box and list form alike).  An LDS-DMA
// instruction with per-lane source addresses costs the wave far more issue time than a global_load_dwordx4 with an SGPR base, and the
// own brick can no longer be requested before the adjacency row is known.  Kept for A/B runs only (variant 15):
//   * every wave works out the links of the whole group from scalars (the list entries, the compact face rows and the link bits of the
//     registered table): no exchange through LDS, no barrier in front of the staging -- one barrier per group instead of two;
//   * the whole own tile (own rows, k- / j-faces, kj-edge rows) arrives by ceil(ELEMS / 128) global_load_lds_dwordx4 instructions with
//     per-lane source addresses (DmaTable): no staging registers, no ds_write for it; only what a wave WITHOUT an i-neighbour sibling
//     needs on that side (i-face columns, i-side edge / corner cells) still travels through registers into the slot on that side;
//   * groups whose runs had to be 2-coloured (two active neighbours that are not i-neighbours) keep the two phases of k_cube_dense:
//     a phase-1 wave requests its tile only after phase 0 is done with the slots that lie inside it.
typedef const __attribute__((address_space(1))) void *dma_src;
typedef __attribute__((address_space(3))) void *dma_dst;

template <int KIND, int NW, int OCC, int PIPE = 1>
__global__ __launch_bounds__(64 * NW, OCC) void k_cube_dma(
    const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out, size_t step,
    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf, const uint32_t *__restrict__ linkmask,
    const uint32_t *__restrict__ faces, const uint32_t *__restrict__ cubemask, uint32_t cs1, uint32_t cs2) {
  constexpr int R = CubeTraits<KIND>::R;
  using TL = DenseTile<KIND>;
  using X = DX<KIND>;
  using DM = DmaX<KIND>;
  constexpr int E = TL::ELEMS;
  __shared__ double lds[(NW + 1) * E];
  __shared__ double jslot[KIND == kCube125 && PIPE == 5 ? NW * 192 : 1];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + (w + 1) * E;
  double *left = tile - E;
  const int roff = w == NW - 1 ? -NW * E : E;
  double *right = tile + roff;

  // descriptors (they depend on the lane only): requested before anything else
  uint32_t dd[DM::NI], xd[X::EXTRA_ITERS];
  {
    const uint4 *dt = reinterpret_cast<const uint4 *>(dma_table<KIND>());
#pragma unroll
    for (int g = 0; g < DM::QUADS; ++g) {
      const uint4 q = dt[g * 64 + lane];
      if (4 * g < DM::NI) dd[4 * g] = q.x;
      if (4 * g + 1 < DM::NI) dd[4 * g + 1] = q.y;
      if (4 * g + 2 < DM::NI) dd[4 * g + 2] = q.z;
      if (4 * g + 3 < DM::NI) dd[4 * g + 3] = q.w;
    }
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
  // ---- the group's bricks and the links between them: scalars, the same in every wave of the group
  uint32_t gb[NW];
#pragma unroll
  for (int v = 0; v < NW; ++v) {
    const uint32_t sl = wg * NW + v;
    gb[v] = sl < n ? __builtin_amdgcn_readfirstlane(ids ? ids[sl] : first + sl) : BRK_LIST_SKIP;
  }
  uint32_t linkbits = 0; // bit v: entries v and v + 1 are i-neighbours that see the same world across their common face
#pragma unroll
  for (int v = 0; v + 1 < NW; ++v)
    if (gb[v] != BRK_LIST_SKIP && gb[v + 1] != BRK_LIST_SKIP) {
      const uint32_t bv = gb[v];
      const uint32_t ip = faces[8 * (size_t)bv + 5];
      const uint32_t rw = linkmask[4 * (size_t)(bv >> 6) + ((bv >> 4) & 2u) + 1];
      if (ip == gb[v + 1] && ((rw >> (bv & 31u)) & 1u)) linkbits |= 1u << v;
    }
  linkbits = __builtin_amdgcn_readfirstlane(linkbits);
  uint32_t bm = 0; // bit v: wave v starts a run that cannot share with the run in front of it
#pragma unroll
  for (int v = 1; v < NW; ++v)
    if (gb[v] != BRK_LIST_SKIP && gb[v - 1] != BRK_LIST_SKIP && !((linkbits >> (v - 1)) & 1u)) bm |= 1u << v;
  const int my_phase = __builtin_popcount(bm & ((2u << w) - 1u)) & 1;
  const int nphase = bm ? 2 : 1;
  uint32_t b = gb[0];
#pragma unroll
  for (int v = 1; v < NW; ++v)
    if (w == (uint32_t)v) b = gb[v];
  const bool active = b != BRK_LIST_SKIP;
  const bool link_l = w > 0 && ((linkbits >> (w - 1)) & 1u), link_r = ((linkbits >> w) & 1u) != 0; // bit NW-1 is never set

  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) d2 *gc2;
  constexpr bool W16 = R == 2;
  double below = 0.0, above = 0.0, fim[R], fip[R], xv[X::EXTRA_ITERS];
  d2 xv2[X::EXTRA_ITERS], im2 = {0.0, 0.0}, ip2 = {0.0, 0.0};
  uint32_t arow = 0;
  auto request_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < DM::NI; ++q) {
      const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((dd[q] >> 16) & 31u) << 2), (int)arow);
      const double *src = in + (size_t)nb * step + ((dd[q] >> 21) & 511u);
      if (q * 128 + 126 < E || q * 128 + lane * 2 < E)
        __builtin_amdgcn_global_load_lds((dma_src)src, (dma_dst)(tile + q * 128), 16, 0, 0);
    }
  };
  if (active) {
    const int l27 = lane < 27 ? lane : 13;
    const bool predicted = cubemask && ((cubemask[b >> 5] >> (b & 31u)) & 1u);
    if (predicted) arow = b + (uint32_t)(l27 / 9 - 1) * cs2 + (uint32_t)((l27 / 3) % 3 - 1) * cs1 + (uint32_t)(l27 % 3 - 1);
    else arow = adj[(size_t)b * 27 + l27];
    if (my_phase == 0) request_tile();
    // the diamond's outermost levels stay in registers (lane = (j, i))
    if (!TL::has_plane(0)) below = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)) * step)[(8 - R) * 64 + lane];
    if (!TL::has_plane(R + 8 + R - 1)) above = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0)) * step)[(R - 1) * 64 + lane];
#pragma unroll
    for (int r = 0; r < R; ++r) fim[r] = fip[r] = 0.0;
    if (!link_l) {
      const gcptr p_im = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1)) * step);
      if (W16) im2 = *(gc2)(p_im + hi * 64 + lo * 8 + 6);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (!W16) fim[r] = p_im[hi * 64 + lo * 8 + (8 - R + r)];
    }
    if (!link_r) {
      const gcptr p_ip = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1)) * step);
      if (W16) ip2 = *(gc2)(p_ip + hi * 64 + lo * 8);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (!W16) fip[r] = p_ip[hi * 64 + lo * 8 + r];
    }
    auto gather = [&](int t) {
      const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((xd[t] >> 16) & 31u) << 2), (int)arow);
      if (X::PAIR) xv2[t] = xd[t] != 0xFFFFFFFFu ? *(const d2 *)(in + (size_t)nb * step + ((xd[t] >> 21) & 511u)) : d2{0.0, 0.0};
      else xv[t] = xd[t] != 0xFFFFFFFFu ? in[(size_t)nb * step + ((xd[t] >> 21) & 511u)] : 0.0;
    };
    if (!link_l) {
#pragma unroll
      for (int t = X::IT0; t < X::IT0 + X::ITS; ++t) gather(t);
    }
    if (!link_r) {
#pragma unroll
      for (int t = X::IT0 + X::ITS; t < X::EXTRA_ITERS; ++t) gather(t);
    }
  }

  gptr dst = (gptr)(out + (size_t)(active ? b : 0) * step);
  asm volatile("" : "+s"(dst));
  for (int ph = 0; ph < nphase; ++ph) {
    const bool mine = active && my_phase == ph;
    if (mine) {
      if (ph == 1) request_tile(); // phase 0 has finished with the slots inside this tile
      if (!link_l) {
        if (W16) *(d2 *)&left[TL::at(hi + R, lo + R, 8 - R)] = im2;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (!W16) left[TL::at(hi + R, lo + R, 8 - R + r)] = fim[r];
#pragma unroll
        for (int t = X::IT0; t < X::IT0 + X::ITS; ++t)
          if (xd[t] != 0xFFFFFFFFu) {
            if (X::PAIR) *(d2 *)&left[xd[t] & 0xFFFFu] = xv2[t];
            else left[xd[t] & 0xFFFFu] = xv[t];
          }
      }
      if (!link_r) {
        if (W16) *(d2 *)&right[TL::at(hi + R, lo + R, 0)] = ip2;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (!W16) right[TL::at(hi + R, lo + R, r)] = fip[r];
#pragma unroll
        for (int t = X::IT0 + X::ITS; t < X::EXTRA_ITERS; ++t)
          if (xd[t] != 0xFFFFFFFFu) {
            if (X::PAIR) *(d2 *)&right[xd[t] & 0xFFFFu] = xv2[t];
            else right[xd[t] & 0xFFFFu] = xv[t];
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tile has landed (LDS-DMA is counted in vmcnt)
    __syncthreads();                                  // the tiles of this phase are complete
    if (mine) {
      int col[2 * R + 1];
#pragma unroll
      for (int di = -R; di <= R; ++di) {
        const int c = lo + di;
        col[di + R] = c < 0 ? -E + c + 8 : (c > 7 ? roff + c - 8 : c);
      }
      if (KIND == kDiamond25) {
        const int co[5] = {col[0], col[1], col[2], col[3], col[4]};
        diamond25_accumulate<TL::T * 8, 8, PIPE == 1>(tile + TL::at(0, hi + R, 0), co, cf, dst, lane, below, above);
      } else if (KIND == kCube125) {
        const int co[5] = {col[0], col[1], col[2], col[3], col[4]};
        cube125_accumulate_jout<TL::T * 8, 8>(tile, co, cf, jslot + (KIND == kCube125 && PIPE == 5 ? w * 192 : 0), (double *)dst, lane);
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
    if (ph + 1 < nphase) __syncthreads(); // phase 1 overwrites slots phase 0 was reading
  }
}

// ---- round 5 experiment: the 125-point kernel at FOUR waves per SIMD (k_cube125_q) --------------------------------------------------
// What a 16-waves-per-CU form of the output-side kernel needs: <= 128 VGPRs and <= 40 960 B of LDS per 4-wave group (k_cube_dense<kCube125,
// 4, 3, 5>: 166 VGPRs, 52 240 B).  This is the LEAN form of the fetch phase -- groups of four fully linked i-neighbours only: no break
// flags, no second phase, the two interior waves gather nothing on their i-sides -- with the shift slots aliased into the wave's own tile
// (cube125_accumulate_jout<..., ALIASED>).  LAYOUT 0 is a TIMING DIAGNOSTIC: the fallback slot F is aliased onto the tiles of the other end
// wave, so the instruction stream, the LDS footprint (36 864 B) and the bank behaviour are those of the real kernel but the halo columns of
// the group's two ends are wrong.  It answers "what is a fourth wave per SIMD worth" before the exact 40 960-byte layout is built.
template <int NW, int OCC, int LAYOUT>
__global__ __launch_bounds__(64 * NW, OCC) void k_cube125_q(const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out,
                                                        size_t step, const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf) {
  constexpr int R = 2;
  using TL = DenseTile<kCube125>;
  using X = DX<kCube125>;
  constexpr int E = TL::ELEMS;
  __shared__ double lds[NW * E];
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) d2 *gc2;
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + wg_in_xcd(cf, per_xcd);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wg * NW + w;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  const int l5 = lane & 31, lp = lane >> 5;
  double *tile = lds + w * E;
  double *left = w == 0 ? lds + (NW - 1) * E : tile - E;   // LAYOUT 0: the left end's slot is the last tile (wrong data, right traffic)
  const int roff = w == NW - 1 ? -(NW - 1) * E : E;
  double *right = tile + roff;
  const uint32_t b = slot < n ? __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot) : BRK_LIST_SKIP;
  if (b == BRK_LIST_SKIP) { // experiment: full groups only
    __syncthreads();
    return;
  }
  const int l27 = lane < 27 ? lane : 13;
  const uint32_t arow = adj[(size_t)b * 27 + l27];
  const gcptr self = sbase(in + (size_t)b * step);
  d2 own2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) own2[q] = *(gc2)(self + q * 128 + 2 * lane);
  __builtin_amdgcn_sched_barrier(0);
  const gcptr p_km = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)) * step);
  const gcptr p_kp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0)) * step);
  const gcptr p_jm = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0)) * step);
  const gcptr p_jp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0)) * step);
  const d2 km2 = *(gc2)(p_km + 6 * 64 + 2 * lane), kp2 = *(gc2)(p_kp + 2 * lane);
  const d2 jm2 = *(gc2)(p_jm + hi * 64 + 48 + 2 * lo), jp2 = *(gc2)(p_jp + hi * 64 + 2 * lo);
  // descriptors of the edge / corner pairs: quad 0 = {own kj-edges, left 0, left 1, right 0}, quad 1 = {right 1}
  const uint4 *xt = reinterpret_cast<const uint4 *>(dense_table<kCube125>());
  static_assert(X::IT0 == 1 && X::ITS == 2, "descriptor layout of the radius-2 cube");
  const uint4 q0 = xt[lane];
  auto gather = [&](uint32_t d) -> d2 {
    const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((d >> 16) & 31u) << 2), (int)arow);
    return d != 0xFFFFFFFFu ? *(const d2 *)(in + (size_t)nb * step + ((d >> 21) & 511u)) : d2{0.0, 0.0};
  };
  const d2 g0 = gather(q0.x);
  // the group's two end waves: the i-face columns and the i-side edge / corner pairs of that side (uniform branch per wave)
  d2 side = {0.0, 0.0}, gs0 = {0.0, 0.0}, gs1 = {0.0, 0.0};
  uint32_t ds0 = 0xFFFFFFFFu, ds1 = 0xFFFFFFFFu;
  if (w == 0) {
    side = *(gc2)(sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1)) * step) + hi * 64 + lo * 8 + 6);
    ds0 = q0.y, ds1 = q0.z;
    gs0 = gather(ds0), gs1 = gather(ds1);
  } else if (w == NW - 1) {
    side = *(gc2)(sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1)) * step) + hi * 64 + lo * 8);
    ds0 = q0.w, ds1 = xt[64 + lane].x;
    gs0 = gather(ds0), gs1 = gather(ds1);
  }
  // ---- stage
#pragma unroll
  for (int q = 0; q < 4; ++q) *(d2 *)&tile[TL::at(2 * q + lp + R, (l5 >> 2) + R, 2 * (l5 & 3))] = own2[q];
  *(d2 *)&tile[TL::at(lp, (l5 >> 2) + R, 2 * (l5 & 3))] = km2;
  *(d2 *)&tile[TL::at(R + 8 + lp, (l5 >> 2) + R, 2 * (l5 & 3))] = kp2;
  *(d2 *)&tile[TL::at(hi + R, 0, 0) + 2 * lo] = jm2;
  *(d2 *)&tile[TL::at(hi + R, R + 8, 0) + 2 * lo] = jp2;
  if (q0.x != 0xFFFFFFFFu) *(d2 *)&tile[q0.x & 0xFFFFu] = g0;
  if (w == 0) {
    *(d2 *)&left[TL::at(hi + R, lo + R, 8 - R)] = side;
    if (ds0 != 0xFFFFFFFFu) *(d2 *)&left[ds0 & 0xFFFFu] = gs0;
    if (ds1 != 0xFFFFFFFFu) *(d2 *)&left[ds1 & 0xFFFFu] = gs1;
  } else if (w == NW - 1) {
    *(d2 *)&right[TL::at(hi + R, lo + R, 0)] = side;
    if (ds0 != 0xFFFFFFFFu) *(d2 *)&right[ds0 & 0xFFFFu] = gs0;
    if (ds1 != 0xFFFFFFFFu) *(d2 *)&right[ds1 & 0xFFFFu] = gs1;
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  gptr dst = (gptr)(out + (size_t)b * step);
  asm volatile("" : "+s"(dst));
  int co[5];
#pragma unroll
  for (int di = -R; di <= R; ++di) {
    const int c = lo + di;
    co[di + R] = c < 0 ? (int)(left - tile) + c + 8 : (c > 7 ? roff + c - 8 : c);
  }
  cube125_accumulate_jout<TL::T * 8, 8, true>(tile, co, cf, tile, (double *)dst, lane);
}

#elif BRK_VARIANTS_PART == 2
// inside launch_cube_dense (BRK_CD, wgs4, x2 are defined there): early returns for the variants
  if (variant == 15 && linkmask && cubereg.faces) { // registered tables: the LDS-DMA staged kernels (slower; see k_cube_dma)
#define BRK_CDMA(KIND, NW, OCC, PIPE)                                                                                                                    \
  do {                                                                                                                                                  \
    brki_note_kernel("k_cube_dma<" #KIND "," #NW "," #OCC "," #PIPE ">");                                                                                \
    hipLaunchKernelGGL((k_cube_dma<KIND, NW, OCC, PIPE>), dim3(wgs4), dim3(64 * NW), g_pad_lds, st, adj, in, out, step, ids, first, n, cf, linkmask, \
                       cubereg.faces, cubereg.mask, cubereg.s1, cubereg.s2);                                                                            \
  } while (0)
    if (x2) BRK_CDMA(kDiamond25, 4, 4, 1);
    else if (stencil_id == BRK_27PT_COEFF) BRK_CDMA(kCube27, 4, 5, 1);
    else BRK_CDMA(kCube125, 4, 3, 5);
#undef BRK_CDMA
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  if (stencil_id == BRK_MPI125 && (variant == 16 || variant == 17)) { // round-5 experiment: the lean 125-point kernel at 4 waves per SIMD in groups of 4 (16) / 8 (17) waves (timing diagnostic, wrong i-ends)
    brki_note_kernel(variant == 16 ? "k_cube125_q<4,4,0>" : "k_cube125_q<8,4,0>");
    if (variant == 16) hipLaunchKernelGGL((k_cube125_q<4, 4, 0>), dim3(wgs4), dim3(256), g_pad_lds, st, adj, in, out, step, ids, first, n, cf);
    else hipLaunchKernelGGL((k_cube125_q<8, 4, 0>), dim3((((n + 7) / 8 + 7) / 8) * 8), dim3(512), g_pad_lds, st, adj, in, out, step, ids, first, n, cf); // 17: 8-wave groups, 2 per CU
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
  const uint32_t wgs8 = (((n + 7) / 8 + 7) / 8) * 8;
  bool done = true;
  if (x2 && variant == 2) BRK_CD(kDiamond25, 4, 4, 1, wgs4, g_pad_lds);                       // plane loop with read-ahead, 8-byte staging (the default until round 4)
  else if (stencil_id == BRK_27PT_COEFF && variant == 5) BRK_CD(kCube27, 8, 4, 1, wgs8, g_pad_lds);
  else if (stencil_id == BRK_MPI125 && variant == 6) BRK_CD(kCube125, 4, 2, 1, wgs4, g_pad_lds);
  else if (stencil_id == BRK_MPI125 && variant == 7) BRK_CD(kCube125, 8, 2, 1, wgs8, g_pad_lds);
  else if (stencil_id == BRK_MPI125 && variant == 11) BRK_CD(kCube125, 4, 3, 4, wgs4, g_pad_lds); // row-pipelined plane loop, 5 rows in flight (spills)
  else if (stencil_id == BRK_MPI125 && variant == 10) BRK_CD(kCube125, 4, 3, 3, wgs4, g_pad_lds); // input-side accumulation, row-pipelined plane loop
  else if (stencil_id == BRK_MPI125 && variant == 9) BRK_CD(kCube125, 4, 3, 0, wgs4, g_pad_lds);
  else if (stencil_id == BRK_MPI125 && variant == 5) BRK_CD(kCube125, 4, 3, 2, wgs4, g_pad_lds);  // fully unrolled plane loop (the round-1 default)
  else done = false;
  if (done) {
    HIP_TRY(hipGetLastError());
    return BRK_OK;
  }
#endif
