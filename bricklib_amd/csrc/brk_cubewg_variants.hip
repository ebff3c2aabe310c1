// brk_cubewg_variants.hip -- cube stencils (27- and 125-point) with workgroup-shared i-halos.
//
// One wave per brick, NW consecutive list entries per workgroup, full (8+2R)^3 LDS tile per wave (CubeTile of
// brick-device.h: k/j faces by structured coalesced loads, edge/corner cells through the descriptor table).  Every
// cell of the tile with di != 0 -- the two i-faces and 16 of the 20 edge/corner blocks -- exists in the i-neighbour's
// tile at the same (k,j) position, 8 columns away.  When the neighbouring list entry is the i-neighbour those cells
// are neither loaded nor staged: lanes on an i edge read them from the sibling wave's tile.  Entries without an
// i-neighbour sibling (row ends, irregular lists) fill their own halo columns as before.
// Accumulation is the scatter form of cube_brick (each tile value is read once per (dj,di)).
#include "brk_device.h"

namespace brklib {
namespace {

template <int KIND, int NW>
__global__ __launch_bounds__(64 * NW, (CubeTraits<KIND>::R == 1 ? 4 : 2)) void k_cube_wg(
    const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out, size_t step,
    const uint32_t *__restrict__ ids, uint32_t first, uint32_t n, Coeffs cf) {
  constexpr int R = CubeTraits<KIND>::R;
  using TL = CubeTile<R>;
  __shared__ double lds[NW * TL::ELEMS];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wg * NW + w;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, lo = lane & 7;
  double *tile = lds + w * TL::ELEMS;
  auto id_of = [&](uint32_t sl) -> uint32_t { return ids ? ids[sl] : first + sl; };

  uint32_t xd[TL::EXTRA_ITERS]; // gather descriptors first, see cube_stage() in brick-device.h
#pragma unroll
  for (int t = 0; t < TL::EXTRA_ITERS; ++t) xd[t] = extra_table<R>()[t * 64 + lane];
  const uint32_t b = slot < n ? __builtin_amdgcn_readfirstlane(id_of(slot)) : BRK_LIST_SKIP;
  const bool active = b != BRK_LIST_SKIP;
  bool share_im = false, share_ip = false;
  if (active) {
    const uint32_t arow = adj[(size_t)b * 27 + (lane < 27 ? lane : 13)]; // lanes 0..26: the adjacency row
    const uint32_t n_im = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, -1));
    const uint32_t n_ip = __builtin_amdgcn_readlane(arow, adj_slot(0, 0, 1));
    const uint32_t b_prev = w > 0 ? __builtin_amdgcn_readfirstlane(id_of(slot - 1)) : 0xFFFFFFFFu;
    const uint32_t b_next = (w + 1 < NW && slot + 1 < n) ? __builtin_amdgcn_readfirstlane(id_of(slot + 1)) : 0xFFFFFFFFu;
    share_im = n_im == b_prev;
    share_ip = n_ip == b_next;
    const gcptr self = sbase(in + (size_t)b * step);
    const gcptr p_km = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(-1, 0, 0)) * step);
    const gcptr p_kp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(1, 0, 0)) * step);
    const gcptr p_jm = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, -1, 0)) * step);
    const gcptr p_jp = sbase(in + (size_t)__builtin_amdgcn_readlane(arow, adj_slot(0, 1, 0)) * step);
    // ---- loads: own brick, k/j faces, i faces only without a sibling, then edge/corner cells via descriptors
    double own[8], fkm[R], fkp[R], fjm[R], fjp[R], fim[R], fip[R];
#pragma unroll
    for (int k = 0; k < 8; ++k) own[k] = self[k * 64 + lane];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      fkm[r] = p_km[(8 - R + r) * 64 + lane];
      fkp[r] = p_kp[r * 64 + lane];
      fjm[r] = p_jm[hi * 64 + (8 - R + r) * 8 + lo];
      fjp[r] = p_jp[hi * 64 + r * 8 + lo];
      fim[r] = fip[r] = 0.0;
    }
    if (!share_im) {
      const gcptr p_im = sbase(in + (size_t)n_im * step);
#pragma unroll
      for (int r = 0; r < R; ++r) fim[r] = p_im[hi * 64 + lo * 8 + (8 - R + r)];
    }
    if (!share_ip) {
      const gcptr p_ip = sbase(in + (size_t)n_ip * step);
#pragma unroll
      for (int r = 0; r < R; ++r) fip[r] = p_ip[hi * 64 + lo * 8 + r];
    }
    double xv[TL::EXTRA_ITERS];
#pragma unroll
    for (int t = 0; t < TL::EXTRA_ITERS; ++t) {
      const uint32_t side = xd[t] >> 30; // 0: di = 0, 1: di = -1, 2: di = +1, 3: padding
      const bool live = side == 0 || (side == 1 && !share_im) || (side == 2 && !share_ip);
      const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((xd[t] >> 16) & 31u) << 2), (int)arow);
      xv[t] = live ? in[(size_t)nb * step + ((xd[t] >> 21) & 511u)] : 0.0;
      if (!live) xd[t] = 0xFFFFFFFFu;
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
    }
    if (!share_im) {
#pragma unroll
      for (int r = 0; r < R; ++r) tile[TL::at(hi + R, lo + R, r)] = fim[r];
    }
    if (!share_ip) {
#pragma unroll
      for (int r = 0; r < R; ++r) tile[TL::at(hi + R, lo + R, R + 8 + r)] = fip[r];
    }
#pragma unroll
    for (int t = 0; t < TL::EXTRA_ITERS; ++t)
      if (xd[t] != 0xFFFFFFFFu) tile[xd[t] & 0xFFFFu] = xv[t];
  }
  __syncthreads(); // sibling tiles are readable
  if (!active) return;
  gptr dst = (gptr)(out + (size_t)b * step);
  asm volatile("" : "+s"(dst));
  // per-lane offset (relative to tile row (kk, hi+R)) of the column lo+di: own tile, or the sibling's tile 8 columns away
  int col[2 * R + 1];
#pragma unroll
  for (int di = -R; di <= R; ++di) {
    const int c = lo + di;
    col[di + R] = (c < 0 && share_im) ? -TL::ELEMS + R + c + 8 : ((c > 7 && share_ip) ? TL::ELEMS + R + c - 8 : R + c);
  }
  if (KIND == kCube125) {
    const int co[5] = {col[0], col[1], col[2], col[3], col[4]};
    cube125_accumulate<TL::T * TL::P, TL::P>(tile + (hi + R) * TL::P, co, cf, (double *)dst, lane);
    return;
  }
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0;
  const int rowbase = (hi + R) * TL::P;
#pragma unroll
  for (int kk = 0; kk < TL::T; ++kk) {
#pragma unroll
    for (int dj = -R; dj <= R; ++dj)
#pragma unroll
      for (int di = -R; di <= R; ++di) {
        const double v = tile[kk * TL::T * TL::P + dj * TL::P + rowbase + col[di + R]];
#pragma unroll
        for (int dk = -R; dk <= R; ++dk) {
          const int k = kk - R - dk; // output plane fed through tap dk
          if (k >= 0 && k < 8) {
            const double wgt = KIND == kCube27 ? cf.c[adj_slot(dk, dj, di)] : cf.c[cls125(dk, dj, di)];
            acc[k] += wgt * v;
          }
        }
      }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) store_result(&dst[k * 64 + lane], acc[k]);
}

} // namespace

int launch_cube_wg(int stencil_id, int nw, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids,
                   uint32_t first, uint32_t n, const Coeffs &cf, hipStream_t st) {
  if (stencil_id == BRK_27PT_COEFF) {
    const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
    hipLaunchKernelGGL((k_cube_wg<kCube27, 8>), dim3(wgs), dim3(512), 0, st, adj, in, out, step, ids, first, n, cf);
  } else if (nw == 2) {
    const uint32_t wgs = (((n + 1) / 2 + 7) / 8) * 8;
    hipLaunchKernelGGL((k_cube_wg<kCube125, 2>), dim3(wgs), dim3(128), 0, st, adj, in, out, step, ids, first, n, cf);
  } else if (nw == 4) {
    const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
    hipLaunchKernelGGL((k_cube_wg<kCube125, 4>), dim3(wgs), dim3(256), 0, st, adj, in, out, step, ids, first, n, cf);
  } else {
    const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
    hipLaunchKernelGGL((k_cube_wg<kCube125, 8>), dim3(wgs), dim3(512), 0, st, adj, in, out, step, ids, first, n, cf);
  }
  HIP_TRY(hipGetLastError());
  return BRK_OK;
}

} // namespace brklib
