// bricksetup.h -- grid/adjacency construction and array <-> brick copies on the host, with the interface of the
// reference's include/bricksetup.h (init_grid :73-90, iter_grid :131-159, copyToBrick :172-200, copyFromBrick :213-221).
// Written fresh: runtime odometers instead of the reference's recursive tag-dispatched templates.
// Device-side equivalents (no host staging): brk_array_to_brick / brk_brick_to_array in brk.h.
#ifndef BRICK_SETUP_H
#define BRICK_SETUP_H

#include <algorithm>
#include <vector>

#include "brick.h"

// Lexicographic grid of brick ids (contiguous dimension first in `dimlist`) and its adjacency list.  A neighbour
// exists iff its FLAT position is inside the grid -- the reference's test (bricksetup.h:44); bricks of the outer
// ring therefore see wrapped neighbours, and out-of-grid entries are 0.
template <unsigned dims>
BrickInfo<dims> init_grid(unsigned *&grid_ptr, const std::vector<long> &dimlist) {
  long size = 1;
  std::vector<long> stride;
  for (long d : dimlist) {
    stride.push_back(size);
    size *= d;
  }
  grid_ptr = (unsigned *)malloc(size * sizeof(unsigned));
  for (long p = 0; p < size; ++p) grid_ptr[p] = (unsigned)p;
  BrickInfo<dims> bInfo((unsigned)size);
  constexpr unsigned nn = static_power<3, dims>::value;
#pragma omp parallel for
  for (long p = 0; p < size; ++p)
    for (unsigned n = 0; n < nn; ++n) {
      long q = p;
      unsigned digits = n;
      for (unsigned d = 0; d < dims; ++d) { // least significant digit = contiguous dimension
        q += ((long)(digits % 3) - 1) * stride[d];
        digits /= 3;
      }
      bInfo.adj[grid_ptr[p]][n] = (q >= 0 && q < size) ? grid_ptr[q] : 0u;
    }
  return bInfo;
}

// Visit, side by side, every element of the iterated region in the brick structure and in the array.
//  dimlist: iterated extent (cells, contiguous first); padding: halo only the array has; ghost: halo both have (skipped).
//  f(bElem &brick_elem, bElem *array_elem)
template <unsigned dims, typename F, typename T, unsigned... BDims>
inline void iter_grid(const std::vector<long> &dimlist, const std::vector<long> &padding, const std::vector<long> &ghost,
                      bElem *arr, unsigned *grid_ptr, Brick<Dim<BDims...>, T> &brick, F f) {
  typedef Brick<Dim<BDims...>, T> BrickT;
  std::vector<long> tile = {BDims...};
  std::reverse(tile.begin(), tile.end()); // contiguous first, like dimlist
  std::vector<long> sA(dims), sB(dims), lo(dims), cnt(dims);
  long a = 1, b = 1, nbricks = 1;
  for (unsigned d = 0; d < dims; ++d) {
    sA[d] = a;
    sB[d] = b;
    a *= dimlist[d] + 2 * (padding[d] + ghost[d]);
    b *= (dimlist[d] + 2 * ghost[d]) / tile[d];
    lo[d] = ghost[d] / tile[d];
    cnt[d] = dimlist[d] / tile[d];
    nbricks *= cnt[d];
  }
  constexpr unsigned mid = static_power<3, dims>::value / 2;
  constexpr unsigned bsize = BrickT::BRICKSIZE;
  // in-brick offset of element e (e enumerates the brick row-major, contiguous first) under the fold
  std::vector<unsigned> inbrick(bsize);
  std::vector<long> aoff(bsize);
  for (unsigned e = 0; e < bsize; ++e) {
    unsigned rem = e, nvec = 0, wvec = 0;
    long off = 0;
    std::vector<unsigned> idx(dims);
    for (unsigned d = 0; d < dims; ++d) {
      idx[d] = rem % tile[d];
      rem /= tile[d];
      off += idx[d] * sA[d];
    }
    const unsigned folds[] = {BrickT::template foldAt<0>(), dims > 1 ? BrickT::template foldAt<(dims > 1 ? 1 : 0)>() : 1,
                              dims > 2 ? BrickT::template foldAt<(dims > 2 ? 2 : 0)>() : 1,
                              dims > 3 ? BrickT::template foldAt<(dims > 3 ? 3 : 0)>() : 1};
    for (unsigned L = 0; L < dims; ++L) { // slowest dimension first
      const unsigned d = dims - 1 - L, D = (unsigned)tile[d], Fd = folds[L];
      nvec = nvec * (D / Fd) + idx[d] / Fd;
      wvec = wvec * Fd + idx[d] % Fd;
    }
    inbrick[e] = nvec * BrickT::VECLEN + wvec;
    aoff[e] = off;
  }
#pragma omp parallel for
  for (long w = 0; w < nbricks; ++w) {
    long rem = w, goff = 0, abase = 0;
    for (unsigned d = 0; d < dims; ++d) {
      const long s = lo[d] + rem % cnt[d];
      rem /= cnt[d];
      goff += s * sB[d];
      abase += (padding[d] + s * tile[d]) * sA[d];
    }
    const unsigned bid = grid_ptr[goff];
    bElem *bp = brick.dat + (size_t)brick.bInfo->adj[bid][mid] * brick.step;
    for (unsigned e = 0; e < bsize; ++e) f(bp[inbrick[e]], arr + abase + aoff[e]);
  }
}

template <unsigned dims, typename T>
inline void copyToBrick(const std::vector<long> &dimlist, const std::vector<long> &padding, const std::vector<long> &ghost,
                        bElem *arr, unsigned *grid_ptr, T &brick) {
  iter_grid<dims>(dimlist, padding, ghost, arr, grid_ptr, brick, [](bElem &b, bElem *a) { b = *a; });
}

template <unsigned dims, typename T>
inline void copyToBrick(const std::vector<long> &dimlist, bElem *arr, unsigned *grid_ptr, T &brick) {
  std::vector<long> zero(dimlist.size(), 0);
  copyToBrick<dims>(dimlist, zero, zero, arr, grid_ptr, brick);
}

template <unsigned dims, typename T>
inline void copyFromBrick(const std::vector<long> &dimlist, const std::vector<long> &padding,
                          const std::vector<long> &ghost, bElem *arr, unsigned *grid_ptr, T &brick) {
  iter_grid<dims>(dimlist, padding, ghost, arr, grid_ptr, brick, [](bElem &b, bElem *a) { *a = b; });
}

#endif // BRICK_SETUP_H
