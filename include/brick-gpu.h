// brick-gpu.h -- moving brick metadata and storage between host and device (reference include/brick-gpu.h:38-99).
// Include brick-hip.h, not this file.  Sizes are size_t here (the reference's `unsigned size` overflows past 4 GiB).
#ifndef BRICK_GPU_FUNCS_H
#define BRICK_GPU_FUNCS_H

#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "brick.h"
#include "brk.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define gpuCheck(x) _gpuCheck(x, #x, __FILE__, __LINE__)

template <typename T>
void _gpuCheck(T e, const char *func, const char *file, const int line) {
  if (e != gpuSuccess) {
    printf("\"%s\" at %d in %s\n\treturned %d\n-> %s\n", func, line, file, (int)e, gpuGetErrorString(e));
    exit(EXIT_FAILURE);
  }
}

// new BrickInfo whose adjacency list lives on the destination side
template <unsigned dims>
BrickInfo<dims> movBrickInfo(BrickInfo<dims> &bInfo, gpuMemcpyKind kind) {
  assert(kind == gpuMemcpyHostToDevice || kind == gpuMemcpyDeviceToHost);
  BrickInfo<dims> ret = bInfo;
  const size_t size = (size_t)bInfo.nbricks * static_power<3, dims>::value * sizeof(unsigned);
  if (kind == gpuMemcpyHostToDevice)
    gpuCheck(gpuMalloc(&ret.adj, size));
  else
    ret.adj = (typename BrickInfo<dims>::adjlist)malloc(size);
  gpuCheck(gpuMemcpy(ret.adj, bInfo.adj, size, kind));
  return ret;
}

// same, and the BrickInfo struct itself is placed on the destination (kernels then follow bIn.bInfo->adj)
template <unsigned dims>
BrickInfo<dims> *movBrickInfoDeep(BrickInfo<dims> &bInfo, gpuMemcpyKind kind) {
  BrickInfo<dims> temp = movBrickInfo(bInfo, kind);
  BrickInfo<dims> *ret;
  if (kind == gpuMemcpyHostToDevice)
    gpuCheck(gpuMalloc(&ret, sizeof(BrickInfo<dims>)));
  else
    ret = (BrickInfo<dims> *)malloc(sizeof(BrickInfo<dims>));
  gpuCheck(gpuMemcpy(ret, &temp, sizeof(BrickInfo<dims>), kind));
  return ret;
}

// OPTIONAL (not in the reference): promise that the DEVICE adjacency list of `bInfo_dev` is not written again until
// brkReleaseAdjacency(bInfo_dev) -- what every driver of the reference does anyway (BrickInfo is filled once by init_grid /
// BrickDecomp::getBrickInfo).  libbrk.so may then keep derived tables next to it (brk_adjacency_register in brk.h: predicted
// face neighbours, compact face rows, link bits) and read fewer adjacency bytes per sweep; results are identical.  Release BEFORE
// gpuFree(bInfo_dev.adj).  Needs libbrk.so (brk.h); a table that is never registered works as before.
template <unsigned dims>
inline void brkRegisterAdjacency(const BrickInfo<dims> &bInfo_dev) {
  static_assert(dims == 3, "the stencil engine is three-dimensional");
  if (brk_adjacency_register(reinterpret_cast<const uint32_t *>(bInfo_dev.adj), bInfo_dev.nbricks) != BRK_OK) {
    printf("brkRegisterAdjacency: %s\n", brk_last_error());
    exit(EXIT_FAILURE);
  }
}
template <unsigned dims>
inline void brkReleaseAdjacency(const BrickInfo<dims> &bInfo_dev) {
  (void)brk_adjacency_release(reinterpret_cast<const uint32_t *>(bInfo_dev.adj));
}

// The pair above as a scope: registers in the constructor; the destructor releases and THEN frees the device table (the order the
// contract asks for), so a driver cannot leave with the table registered or leaked -- also on an exception path.
template <unsigned dims>
struct brkAdjacencyScope {
  BrickInfo<dims> &bInfo_dev;
  explicit brkAdjacencyScope(BrickInfo<dims> &info_dev) : bInfo_dev(info_dev) { brkRegisterAdjacency(bInfo_dev); }
  brkAdjacencyScope(const brkAdjacencyScope &) = delete;
  brkAdjacencyScope &operator=(const brkAdjacencyScope &) = delete;
  ~brkAdjacencyScope() {
    brkReleaseAdjacency(bInfo_dev);
    (void)gpuFree(bInfo_dev.adj);
    bInfo_dev.adj = nullptr;
  }
};

// OPTIONAL (not in the reference): direction of successive sweeps over the same bricks.  `brkSweepTurn t; ... brk_stencil_apply_box_ex(...,
// t.flags(), stream)` inside the timed lambda walks the bricks forwards, backwards, forwards, ...: a sweep then starts in the part of
// the field the Infinity Cache still holds from the one before (brk.h, BRK_SWEEP_REVERSE; results do not depend on it).
struct brkSweepTurn {
  unsigned turn = 0;
  uint32_t flags() { return (turn++ & 1u) ? BRK_SWEEP_REVERSE : 0u; }
};

// new BrickStorage on the destination side holding a copy of the data.  The device side comes from brk_storage_alloc (brk.h): the
// storage is PLACED in HBM where streams between it and the storage moved before it run fastest (a sweep between two storages in
// DRAM regions of the same kind loses 3 %), which the reference's one-cudaMalloc-per-storage leaves to chance.
inline BrickStorage movBrickStorage(BrickStorage &bStorage, gpuMemcpyKind kind) {
  assert(kind == gpuMemcpyHostToDevice || kind == gpuMemcpyDeviceToHost);
  BrickStorage ret = bStorage;
  const size_t size = bStorage.step * (size_t)bStorage.chunks * sizeof(bElem);
  bElem *datptr;
  if (kind == gpuMemcpyHostToDevice) {
    void *placed = nullptr;
    if (brk_storage_alloc(size, &placed, nullptr) != BRK_OK) {
      printf("brk_storage_alloc: %s\n", brk_last_error());
      exit(EXIT_FAILURE);
    }
    datptr = static_cast<bElem *>(placed);
    gpuCheck(gpuMemcpy(datptr, bStorage.dat.get(), size, kind));
    ret.dat = std::shared_ptr<bElem>(datptr, [](bElem *p) { (void)brk_storage_free(p); });
  } else {
    datptr = (bElem *)malloc(size);
    gpuCheck(gpuMemcpy(datptr, bStorage.dat.get(), size, kind));
    ret.dat = std::shared_ptr<bElem>(datptr, [](bElem *p) { free(p); });
  }
  return ret;
}

// OPTIONAL (not in the reference): the LAST device-related line of a driver's main(), after its own gpuFree calls.  Ends the process's
// use of the GPU in a fixed order on the calling thread instead of leaving it to exit handlers that run beside the runtime's own
// threads (an exit-time SIGSEGV of examples/strong_hip in a thread of the runtime, seen only in drivers that
// are linked against OpenMP and leave main within its workers' spin-wait):
//   1. the OpenMP runtime is shut down (workers joined) -- they otherwise spin for KMP_BLOCKTIME = 200 ms after the last parallel
//      region, on every core, while the process exits;
//   2. brk_finalize(): everything the library enqueued has finished, its device-side caches are released here;
//   3. the device is idle when exit() starts running atexit handlers (the fat-binary unregistration of the executable and of
//      libbrk.so among them).
inline void brkFinalize() {
#ifdef _OPENMP
  (void)omp_pause_resource_all(omp_pause_hard);
#endif
  if (brk_finalize() != BRK_OK) printf("brk_finalize: %s\n", brk_last_error());
  (void)gpuDeviceSynchronize();
}

#endif // BRICK_GPU_FUNCS_H
