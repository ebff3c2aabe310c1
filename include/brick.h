// brick.h -- BrickStorage / BrickInfo / Brick: the data structures of the brick layout, with the public names,
// members and semantics of the reference's include/brick.h (:53-82 BrickStorage, :96-127 BrickInfo, :129-190 Dim /
// cal_size / cal_offs, :214-327 element access, :353-395 Brick).  Written fresh for the MI355X engine; the element
// accessor is one template instead of three specialisations.
#ifndef BRICK_H
#define BRICK_H

#include <stdlib.h>

#include <memory>
#include <stdexcept>
#include <type_traits>

#include "vecscatter.h"

#define ALIGN 2048 // allocation alignment of BrickStorage::allocate

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define FORCUDA __host__ __device__
#else
#define FORCUDA
#endif

template <unsigned base, unsigned exp>
struct static_power {
  static constexpr unsigned value = base * static_power<base, exp - 1>::value;
};
template <unsigned base>
struct static_power<base, 0> {
  static constexpr unsigned value = 1;
};

// Flat storage of `chunks` chunks of `step` elements; several fields may be interleaved in one chunk.
struct BrickStorage {
  std::shared_ptr<bElem> dat;
  long chunks;
  size_t step;
  void *mmap_info = nullptr; // kept for source compatibility; the GPU engine has no mmap views

  static BrickStorage allocate(long chunks, size_t step) {
    BrickStorage s;
    s.chunks = chunks;
    s.step = step;
    size_t bytes = ((size_t)chunks * step * sizeof(bElem) + ALIGN - 1) / ALIGN * ALIGN;
    s.dat = std::shared_ptr<bElem>(static_cast<bElem *>(aligned_alloc(ALIGN, bytes)), [](bElem *p) { free(p); });
    return s;
  }

  // The reference's memfd/mmap allocators (include/brick.h:78-81, src/memfd.cpp) build ALIASED host mappings so that one
  // MPI message covers a neighbour's ghost regions.  That is a host virtual-memory device; on the GPU the same effect comes
  // from the region-ordered numbering (every run is contiguous) plus pack/unpack launches (brk_exchange_*).  The names exist
  // so that code calling them fails at run time with an explanation instead of at link time.
  static BrickStorage mmap_alloc(long, long) {
    throw std::runtime_error("BrickStorage::mmap_alloc: mmap views are not part of the MI355X brick engine; allocate with "
                             "BrickStorage::allocate / movBrickStorage and exchange with BrickDecomp::exchangeView (brk_exchange_*)");
  }
  static BrickStorage mmap_alloc(long chunks, long step, void *, size_t) { return mmap_alloc(chunks, step); }
};

// Adjacency list: adj[b][3^dims]; entry (dk+1)*9+(dj+1)*3+(di+1) for dims == 3, centre = self.
template <unsigned dims>
struct BrickInfo {
  typedef unsigned (*adjlist)[static_power<3, dims>::value];
  adjlist adj;
  unsigned nbricks;

  explicit BrickInfo(unsigned nbricks) : nbricks(nbricks) {
    adj = (adjlist)malloc((size_t)nbricks * static_power<3, dims>::value * sizeof(unsigned));
  }
  BrickStorage allocate(long step) { return BrickStorage::allocate(nbricks, step); }
};

template <unsigned... Ds>
struct Dim {};

template <unsigned... xs>
struct cal_size;
template <unsigned x>
struct cal_size<x> {
  static constexpr unsigned value = x;
};
template <unsigned x, unsigned... xs>
struct cal_size<x, xs...> {
  static constexpr unsigned value = x * cal_size<xs...>::value;
};

// offset of a neighbour (digits in {0,1,2}, slowest dimension first) inside an adjacency row
template <unsigned... offs>
struct cal_offs;
template <unsigned off>
struct cal_offs<1, off> {
  static constexpr unsigned value = off;
};
template <unsigned dim, unsigned off, unsigned... offs>
struct cal_offs<dim, off, offs...> {
  static constexpr unsigned value = off * static_power<3, dim - 1>::value + cal_offs<dim - 1, offs...>::value;
};

namespace brk_detail {
// (identifiers here avoid the short upper-case names drivers use as macros: N, TILE, GZ, ...)
template <unsigned idx_, unsigned... xs>
struct nth;
template <unsigned x, unsigned... xs>
struct nth<0, x, xs...> {
  static constexpr unsigned value = x;
};
template <unsigned idx_, unsigned x, unsigned... xs>
struct nth<idx_, x, xs...> {
  static constexpr unsigned value = nth<idx_ - 1, xs...>::value;
};
} // namespace brk_detail

template <typename...>
struct Brick;

// Element access b[brick][k][j][i] with every index in [-D, 2D): an index outside [0,D) moves to the neighbouring
// brick in that dimension.  In-brick position follows the vector folding of the reference (brick.h:235-246): the last
// sizeof...(Folds) dimensions are folded; an unfolded dimension behaves like fold 1.
template <typename BrickT, unsigned Level>
struct _BrickAccessor;

template <unsigned... BDims, unsigned... Folds>
struct Brick<Dim<BDims...>, Dim<Folds...>> {
  typedef Brick<Dim<BDims...>, Dim<Folds...>> mytype;
  typedef BrickInfo<sizeof...(BDims)> myBrickInfo;
  static constexpr unsigned NDIMS = sizeof...(BDims);
  static constexpr unsigned NFOLDS = sizeof...(Folds);
  static constexpr unsigned VECLEN = cal_size<Folds...>::value;
  static constexpr unsigned BRICKSIZE = cal_size<BDims...>::value;

  myBrickInfo *bInfo;
  size_t step;
  bElem *dat;
  BrickStorage bStorage;

  template <unsigned L>
  static constexpr unsigned dimAt() {
    return brk_detail::nth<L, BDims...>::value;
  }
  template <unsigned L>
  static constexpr unsigned foldAt() {
    return L + NFOLDS >= NDIMS ? brk_detail::nth<(L + NFOLDS >= NDIMS ? L + NFOLDS - NDIMS : 0), Folds...>::value : 1;
  }

  FORCUDA inline _BrickAccessor<mytype, 0> operator[](unsigned b) { return _BrickAccessor<mytype, 0>{this, b, 0, 0, 0}; }

  template <unsigned... Offsets>
  FORCUDA inline bElem *neighbor(unsigned b) {
    return &dat[bInfo->adj[b][cal_offs<sizeof...(BDims), Offsets...>::value] * step];
  }

  Brick(myBrickInfo *bInfo, const BrickStorage &brickStorage, unsigned offset) : bInfo(bInfo) {
    bStorage = brickStorage;
    dat = bStorage.dat.get() + offset;
    step = (unsigned)bStorage.step;
  }
};

template <typename BrickT, unsigned Level>
struct _BrickAccessor {
  BrickT *par;
  unsigned b, pos, nvec, wvec;

  FORCUDA inline auto operator[](unsigned i) ->
      typename std::conditional<Level + 1 == BrickT::NDIMS, bElem &, _BrickAccessor<BrickT, Level + 1>>::type {
    return step(i, std::integral_constant<bool, Level + 1 == BrickT::NDIMS>());
  }

 private:
  struct Next {
    unsigned pos, nvec, wvec;
  };
  FORCUDA inline Next advance(unsigned i) const {
    constexpr unsigned D = BrickT::template dimAt<Level>();
    constexpr unsigned F = BrickT::template foldAt<Level>();
    const unsigned shifted = i + D; // [-D,2D) -> [0,3D)
    const unsigned l = shifted % D;
    return Next{pos * 3 + shifted / D, nvec * (D / F) + l / F, wvec * F + l % F};
  }
  FORCUDA inline bElem &step(unsigned i, std::true_type) {
    const Next n = advance(i);
    return par->dat[par->bInfo->adj[b][n.pos] * par->step + n.nvec * BrickT::VECLEN + n.wvec];
  }
  FORCUDA inline _BrickAccessor<BrickT, Level + 1> step(unsigned i, std::false_type) {
    const Next n = advance(i);
    return _BrickAccessor<BrickT, Level + 1>{par, b, n.pos, n.nvec, n.wvec};
  }
};

#endif // BRICK_H
