// dev_shl.h -- the lane rotation primitive of bricklib's generated GPU code (reference include/dev_shl.h:48-63), for
// hand-written kernels that still use it.  Two registers l and r hold consecutive vectors of `cw` lanes each; the
// result is the window that starts `cw - kn` lanes into l: lanes with in-vector index cid < kn read l from cw - kn lanes
// further up, the others read r from kn lanes further down.  The kernels of this library do not use it: they share
// neighbouring values through LDS tiles (brick-device.h) instead of rotating registers.
#pragma once

#if defined(__HIP__)
#include <hip/hip_runtime.h>

template <typename T>
__device__ __forceinline__ void dev_shl(T &res, T l, T r, int kn, int cw, int cid) {
  const int lane = (int)__lane_id();
  const T upper = __shfl(l, lane + (cw - kn)); // ds_bpermute with absolute source lanes
  const T lower = __shfl(r, lane - kn);
  res = cid < kn ? upper : lower;
}
#endif
