// multiarray.h -- plain padded arrays used next to bricks by the drivers (reference include/multiarray.h,
// src/multiarray.cpp).  Header-only here.  randomArray is SEEDED (counter hash of the element index) unlike the
// reference's random_device (src/multiarray.cpp:13-23), so runs are reproducible; call brkSeed() to change it.
#ifndef MULTIARRAY_H
#define MULTIARRAY_H

#include <cmath>
#include <cstdint>
#include <vector>

#include "brick.h"
#include "cmpconst.h"

inline uint64_t &brkSeedRef() {
  static uint64_t seed = 42;
  return seed;
}
inline void brkSeed(uint64_t s) { brkSeedRef() = s; }

inline bElem *uninitArray(const std::vector<long> &list, long &size) {
  size = 1;
  for (long i : list) size *= i;
  size_t bytes = ((size_t)size * sizeof(bElem) + ALIGN - 1) / ALIGN * ALIGN;
  return (bElem *)aligned_alloc(ALIGN, bytes);
}

inline bElem *randomArray(const std::vector<long> &list) {
  long size;
  bElem *arr = uninitArray(list, size);
  const uint64_t seed = brkSeedRef();
#pragma omp parallel for
  for (long l = 0; l < size; ++l) {
    uint64_t x = seed * 0xD1342543DE82EF95ull + (uint64_t)l + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    arr[l] = (bElem)((double)(x >> 11) * (1.0 / 9007199254740992.0)); // U[0,1)
  }
  return arr;
}

inline bElem *zeroArray(const std::vector<long> &list) {
  long size;
  bElem *arr = uninitArray(list, size);
#pragma omp parallel for
  for (long l = 0; l < size; ++l) arr[l] = 0.0;
  return arr;
}

inline bool compareArray(const std::vector<long> &list, bElem *arrA, bElem *arrB) {
  long size = 1;
  for (long i : list) size *= i;
  bool same = true;
#pragma omp parallel for reduction(&& : same)
  for (long l = 0; l < size; ++l) {
    const bElem diff = std::abs(arrA[l] - arrB[l]);
    same = same && ((diff < BRICK_TOLERANCE) || (diff < (std::abs(arrA[l]) + std::abs(arrB[l])) * BRICK_TOLERANCE));
  }
  return same;
}

#endif // MULTIARRAY_H
