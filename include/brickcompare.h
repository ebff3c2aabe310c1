// brickcompare.h -- compareBrick: element-wise comparison of bricks against an array over the iterated region, with the
// acceptance rule of the reference (include/brickcompare.h:35-39): |a-b| < tol  or  |a-b| < (|a|+|b|)*tol.
// Device-side equivalent: brk_compare_brick in brk.h.
#ifndef BRICK_BRICKCOMPARE_H
#define BRICK_BRICKCOMPARE_H

#include <atomic>
#include <cmath>

#include "bricksetup.h"
#include "cmpconst.h"

template <unsigned dims, typename T>
inline bool compareBrick(const std::vector<long> &dimlist, const std::vector<long> &padding, const std::vector<long> &ghost,
                         bElem *arr, unsigned *grid_ptr, T &brick) {
  std::atomic<bool> same(true);
  iter_grid<dims>(dimlist, padding, ghost, arr, grid_ptr, brick, [&same](bElem &b, const bElem *a) {
    const double diff = std::abs(b - *a);
    if (!((diff < BRICK_TOLERANCE) || (diff < (std::abs(b) + std::abs(*a)) * BRICK_TOLERANCE)))
      same.store(false, std::memory_order_relaxed);
  });
  return same.load();
}

template <unsigned dims, typename T>
inline bool compareBrick(const std::vector<long> &dimlist, bElem *arr, unsigned *grid_ptr, T &brick) {
  std::vector<long> zero(dimlist.size(), 0);
  return compareBrick<dims, T>(dimlist, zero, zero, arr, grid_ptr, brick);
}

#endif // BRICK_BRICKCOMPARE_H
