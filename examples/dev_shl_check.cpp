// Checks include/dev_shl.h on the GPU against its definition: lanes with in-vector index cid < kn read l from cw - kn lanes
// up, the others read r from kn lanes down (vector widths 8 and 64, several shifts).
#include <brick-hip.h>

#include <cstdio>

__global__ void rotate(const double *l, const double *r, double *out, int kn, int cw) {
  double res;
  dev_shl(res, l[threadIdx.x], r[threadIdx.x], kn, cw, (int)(threadIdx.x & (cw - 1)));
  out[threadIdx.x] = res;
}

int main() {
  double hl[64], hr[64], ho[64], *l, *r, *o;
  for (int i = 0; i < 64; ++i) hl[i] = i, hr[i] = 100 + i;
  gpuCheck(gpuMalloc(&l, sizeof hl));
  gpuCheck(gpuMalloc(&r, sizeof hr));
  gpuCheck(gpuMalloc(&o, sizeof ho));
  gpuCheck(gpuMemcpy(l, hl, sizeof hl, gpuMemcpyHostToDevice));
  gpuCheck(gpuMemcpy(r, hr, sizeof hr, gpuMemcpyHostToDevice));
  int bad = 0;
  for (int cw : {8, 64})
    for (int kn : {1, 3, 7}) {
      gpuExecKernel(rotate, dim3(1), dim3(64), l, r, o, kn, cw);
      gpuCheck(gpuMemcpy(ho, o, sizeof ho, gpuMemcpyDeviceToHost));
      for (int i = 0; i < 64; ++i) {
        const int cid = i & (cw - 1);
        if (cid < kn ? (i + cw - kn < 64 && ho[i] != hl[i + cw - kn]) : ho[i] != hr[i - kn]) ++bad;
      }
    }
  printf(bad ? "dev_shl: %d mismatches\n" : "dev_shl: ok\n", bad);
  return bad != 0;
}
