// pfprobe -- development probe (not part of the product): can the SCALAR memory path pull bricks into the L2 / Infinity Cache ahead
// of the vector loads that need them?  gfx950 has no prefetch instruction; an s_load_dword whose result is never read is one: it
// travels scalar cache -> L2 -> fabric, outside the vector L1 and its bounded number of requests in flight, and leaves the 128-byte
// line in the XCD's L2.  The probe is the brick-shaped copy of tools/l1probe (one wave per 4-KiB brick, XCD-slab order, NH k-planes
// from the bricks below / above, non-temporal stores); each wave additionally touches the brick `dist` list slots ahead in its XCD's
// share, one s_load_dword per `gran` bytes.
//   pfprobe [-n N] [-i iters] [-w waves_per_wg 4|8] [-l extra_lds_bytes]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                \
  do {                                                                       \
    hipError_t e = (x);                                                      \
    if (e != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
      exit(2);                                                               \
    }                                                                        \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) char *gbytes;
typedef const __attribute__((address_space(4))) char *cbytes;

__global__ void k_fill(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)(i % 977) * 1e-3;
}

// one s_load_dword per line of the brick at `base`; nothing ever reads the result.  The loads land LATER than the asm statement the
// compiler sees: they all write s101, which the clobber list keeps away from the register allocator for the whole kernel (the first
// version let the compiler pick the register, reused it, and a late load overwrote a live pointer: a memory fault on the GPU box)
template <int LINES, int GRAN>
__device__ __forceinline__ void touch_brick(const double *base) {
#pragma unroll
  for (int q = 0; q < LINES; ++q) asm volatile("s_load_dword s101, %0, %1" ::"s"(base), "n"(q * GRAN) : "s101");
}

// NH: k-planes taken from each of the two k-neighbours; NW waves per workgroup; PF: 0 none, 1 = one touch per 128 B, 2 = one per 64 B
// WHO: which waves of a workgroup touch (1 = every wave its own slot + dist; 2 = wave 0 for the whole group)
template <int NH, int NW, int PF>
__global__ __launch_bounds__(64 * NW) void k_probe(const double *__restrict__ in, double *__restrict__ out, const uint32_t *__restrict__ ids,
                                                    uint32_t n, uint32_t kstride, uint32_t dist, int halo_too) {
  extern __shared__ double dyn[];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t local = blockIdx.x >> 3, x = blockIdx.x & 7u;
  const uint32_t wg = x * per_xcd + local;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slot = wg * NW + w;
  if (slot >= n) return;
  const int lane = threadIdx.x & 63;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids[slot]);
  uint32_t pb = 0xFFFFFFFFu;
  if (PF) {
    const uint32_t ahead = local * NW + w + dist; // position inside this XCD's share
    if (ahead < per_xcd * NW && slot + dist < n) pb = __builtin_amdgcn_readfirstlane(ids[slot + dist]);
  }
  gbytes self = (gbytes)(in + (size_t)b * 512);
  gbytes below = (gbytes)(in + (size_t)(b - kstride) * 512 + (8 - NH) * 64);
  gbytes above = (gbytes)(in + (size_t)(b + kstride) * 512);
  asm volatile("" : "+s"(self), "+s"(below), "+s"(above));
  d2 o[4], h[NH > 0 ? NH : 1];
  const uint32_t v16 = lane * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(o[q]) : "v"(v16 + q * 1024), "s"(self));
  if (NH > 0) {
#pragma unroll
    for (int q = 0; q < NH / 2; ++q) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(h[q]) : "v"(v16 + q * 1024), "s"(below));
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(h[NH / 2 + q]) : "v"(v16 + q * 1024), "s"(above));
    }
  }
  // the touches go out BEHIND the wave's own requests
  if (PF && pb != 0xFFFFFFFFu) {
    const double *p = in + (size_t)pb * 512;
    if (PF == 1) touch_brick<32, 128>(p);
    else touch_brick<64, 64>(p);
    if (halo_too && NH > 0) { // the k-neighbour planes of that brick as well (lines some other wave will own later anyway)
      const double *pl = in + (size_t)(pb - kstride) * 512 + (8 - NH) * 64, *pu = in + (size_t)(pb + kstride) * 512;
      touch_brick<(NH > 0 ? NH * 4 : 1), 128>(pl);
      touch_brick<(NH > 0 ? NH * 4 : 1), 128>(pu);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
  if (NH > 0) {
#pragma unroll
    for (int q = 0; q < NH; ++q) asm volatile("" : "+v"(h[q]));
  }
  double r[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) r[2 * q] = o[q].x, r[2 * q + 1] = o[q].y;
  if (NH > 0) {
#pragma unroll
    for (int q = 0; q < NH; ++q) r[(2 * q) & 7] += h[q].x, r[(2 * q + 1) & 7] += h[q].y;
  }
  double *dst = out + (size_t)b * 512;
#pragma unroll
  for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(r[k] * 1.0001, &dst[k * 64 + lane]);
  if (dyn[0] == 12345.678 && lane == 77) dst[0] = dyn[1]; // keeps the dynamic LDS allocation alive
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // no touch outlives its wave
}

template <int NH, int PF>
static float run(const double *in, double *out, const uint32_t *ids, uint32_t n, uint32_t kstride, int iters, int nw, size_t extra, uint32_t dist,
                 int halo_too) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto go = [&]() {
    if (nw == 8) {
      const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
      hipLaunchKernelGGL((k_probe<NH, 8, PF>), dim3(wgs), dim3(512), extra, 0, in, out, ids, n, kstride, dist, halo_too);
    } else {
      const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
      hipLaunchKernelGGL((k_probe<NH, 4, PF>), dim3(wgs), dim3(256), extra, 0, in, out, ids, n, kstride, dist, halo_too);
    }
  };
  for (int q = 0; q < 5; ++q) go();
  CK(hipEventRecord(e0));
  for (int q = 0; q < iters; ++q) go();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / iters;
}

int main(int argc, char **argv) {
  int N = 512, iters = 40, nw = 4;
  size_t extra = 0;
  for (int a = 1; a < argc; ++a) {
    std::string o = argv[a];
    if (o == "-n") N = atoi(argv[++a]);
    else if (o == "-i") iters = atoi(argv[++a]);
    else if (o == "-w") nw = atoi(argv[++a]);
    else if (o == "-l") extra = (size_t)atol(argv[++a]);
  }
  const long SB = N / 8 + 2, NB = SB - 2;
  const size_t nbricks = (size_t)SB * SB * SB;
  std::vector<uint32_t> ids;
  const long rows = NB / 8;
  for (long x = 0; x < 8; ++x) // the order of bricklib_amd/order.py: one slab of j-rows per XCD, blocks of 8 rows over all k, i fastest
    for (long jb = 0; jb < rows; jb += 8)
      for (long k = 1; k < SB - 1; ++k)
        for (long j = x * rows + jb; j < x * rows + jb + 8 && j < (x + 1) * rows; ++j)
          for (long i = 1; i < SB - 1; ++i) ids.push_back((uint32_t)((k * SB + (j + 1)) * SB + i));
  double *in, *out;
  uint32_t *ids_d;
  CK(hipMalloc(&in, nbricks * 4096));
  CK(hipMalloc(&out, nbricks * 4096));
  CK(hipMalloc(&ids_d, ids.size() * 4));
  CK(hipMemcpy(ids_d, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  k_fill<<<4096, 256>>>(in, nbricks * 512);
  CK(hipDeviceSynchronize());
  const uint32_t n = (uint32_t)ids.size(), ks = (uint32_t)(SB * SB);
  const double cells = (double)N * N * N;
  printf("# pfprobe N=%d bricks=%u waves/wg=%d extra LDS=%zu B, %d launches each; GB/s = 16 B x cells / time (algorithmic)\n", N, n, nw, extra, iters);
  const uint32_t dists[] = {64, 256, 512, 1024, 2048, 4096, 8192};
#define ROW(NHv, PFv, D, H)                                                                                                              \
  {                                                                                                                                      \
    const float ms = run<NHv, PFv>(in, out, ids_d, n, ks, iters, nw, extra, D, H);                                                       \
    printf("planes/side %d  touch %-9s dist %5u slots%s: %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", NHv,                                  \
           PFv == 0 ? "none" : (PFv == 1 ? "per 128 B" : "per 64 B"), PFv ? D : 0u, H ? " (+halo planes)" : "", ms, 16.0 * cells / ms / 1e6, \
           16.0 * cells / ms / 1e6 / 8000.0);                                                                                            \
    fflush(stdout);                                                                                                                      \
  }
  for (int rep = 0; rep < 2; ++rep) {
    ROW(0, 0, 0, 0)
    for (uint32_t d : dists) ROW(0, 1, d, 0)
    ROW(0, 2, 1024, 0)
    ROW(4, 0, 0, 0)
    for (uint32_t d : dists) ROW(4, 1, d, 0)
    ROW(8, 0, 0, 0)
    for (uint32_t d : dists) ROW(8, 1, d, 0)
  }
  return 0;
}
