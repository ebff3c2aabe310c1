// l1probe -- development probe (not part of the product): what does a byte requested through the vector L1 cost a brick sweep,
// and does the load flavour change it?  One wave per 4-KiB brick in the XCD-slab order of the stencil kernels; every wave reads
// its own brick, NH k-planes of the bricks below and above (lines a neighbour workgroup also reads: L2 hits in the steady
// state), adds everything up and writes 4 KiB with non-temporal stores.  Flavours of the own / halo loads:
//   0 global_load_dwordx2 (8 B per lane)   1 global_load_dwordx4   2 dwordx4 sc1   3 dwordx4 nt   4 dwordx4 sc0 sc1
//   5 global_load_lds_dwordx4 (LDS-DMA, then ds_read_b128)
//   l1probe [-n N] [-i iters] [-w waves_per_wg] [-l extra_lds_bytes]
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

template <int FL>
__device__ __forceinline__ void ld16(d2 &v, gbytes base, uint32_t voff) {
  if (FL == 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(voff), "s"(base));
  else if (FL == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(v) : "v"(voff), "s"(base));
  else if (FL == 3) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(v) : "v"(voff), "s"(base));
  else asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=&v"(v) : "v"(voff), "s"(base));
}

__global__ void k_fill(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)(i % 977) * 1e-3;
}

// OWN / HALO: load flavour; NH: k-planes taken from each of the two k-neighbours (0, 2, 4, 8); NW waves per workgroup
template <int OWN, int HALO, int NH, int NW>
__global__ __launch_bounds__(64 * NW) void k_probe(const double *__restrict__ in, double *__restrict__ out, const uint32_t *__restrict__ ids,
                                                    uint32_t n, uint32_t kstride) {
  extern __shared__ double dyn[];
  __shared__ __attribute__((aligned(16))) double tile[NW * (512 + 2 * 512)];
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t w = threadIdx.x >> 6, slot = wg * NW + w;
  if (slot >= n) return;
  const int lane = threadIdx.x & 63;
  const uint32_t b = __builtin_amdgcn_readfirstlane(ids[slot]);
  gbytes self = (gbytes)(in + (size_t)b * 512);
  gbytes below = (gbytes)(in + (size_t)(b - kstride) * 512 + (8 - NH) * 64);
  gbytes above = (gbytes)(in + (size_t)(b + kstride) * 512);
  asm volatile("" : "+s"(self), "+s"(below), "+s"(above));
  double *my = tile + w * 1536;
  d2 o[4], h[NH > 0 ? NH : 1];
  double o8[8], h8[NH > 0 ? 2 * NH : 1];
  const uint32_t v16 = lane * 16, v8 = lane * 8;
  // ---- own brick
  if (OWN == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) o8[k] = ((const __attribute__((address_space(1))) double *)self)[k * 64 + lane];
  } else if (OWN == 5) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(self + q * 1024 + v16),
                                       (__attribute__((address_space(3))) void *)(my + q * 128), 16, 0, 0);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) ld16<OWN>(o[q], self + q * 1024, v16);
  }
  // ---- k-halo: NH planes below (the neighbour's last planes), NH planes above (its first planes)
  if (NH > 0) {
    if (HALO == 0) {
#pragma unroll
      for (int p = 0; p < NH; ++p) {
        h8[p] = ((const __attribute__((address_space(1))) double *)below)[p * 64 + lane];
        h8[NH + p] = ((const __attribute__((address_space(1))) double *)above)[p * 64 + lane];
      }
    } else if (HALO == 5) {
#pragma unroll
      for (int q = 0; q < NH / 2; ++q) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(below + q * 1024 + v16),
                                         (__attribute__((address_space(3))) void *)(my + 512 + q * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(above + q * 1024 + v16),
                                         (__attribute__((address_space(3))) void *)(my + 1024 + q * 128), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NH / 2; ++q) {
        ld16<HALO>(h[q], below + q * 1024, v16);
        ld16<HALO>(h[NH / 2 + q], above + q * 1024, v16);
      }
    }
  }
  (void)v8;
  // every asm load must be back before its register is read; the compiler's own loads are tracked by the compiler
  if (OWN >= 1 && OWN <= 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
  if (NH > 0 && HALO >= 1 && HALO <= 4) {
#pragma unroll
    for (int q = 0; q < NH; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[q]));
  }
  if (OWN == 5 || (NH > 0 && HALO == 5)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // ---- reduce to 8 values per lane in the (plane, lane) layout of the stencil kernels' stores
  double r[8];
  if (OWN == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = o8[k];
  } else {
    if (OWN == 5) {
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = *(d2 *)&my[q * 128 + 2 * lane];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) r[2 * q] = o[q].x, r[2 * q + 1] = o[q].y;
  }
  if (NH > 0) {
    if (HALO == 0) {
#pragma unroll
      for (int p = 0; p < 2 * NH; ++p) r[p & 7] += h8[p];
    } else {
      if (HALO == 5) {
#pragma unroll
        for (int q = 0; q < NH / 2; ++q) h[q] = *(d2 *)&my[512 + q * 128 + 2 * lane], h[NH / 2 + q] = *(d2 *)&my[1024 + q * 128 + 2 * lane];
      }
#pragma unroll
      for (int q = 0; q < NH; ++q) r[(2 * q) & 7] += h[q].x, r[(2 * q + 1) & 7] += h[q].y;
    }
  }
  double *dst = out + (size_t)b * 512;
#pragma unroll
  for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(r[k] * 1.0001, &dst[k * 64 + lane]);
  if (dyn[0] == 12345.678 && lane == 77) dst[0] = dyn[1]; // keeps the dynamic LDS allocation alive
}


// ---- persistent, software-pipelined form: a wave walks over bricks; the LDS-DMA loads of brick it+D-1 are issued before brick it is
// waited for (counted vmcnt: the stores of the bricks in between are younger than the awaited loads), read from LDS and stored.
// No register ever holds a load in flight.  NH planes per side come from the k-neighbours as above.  R = workgroups per CU.
__device__ __forceinline__ void dma16(uint32_t lds_addr, gbytes base, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NH, int D, int NW>
__global__ __launch_bounds__(64 * NW) void k_pipe(const double *__restrict__ in, double *__restrict__ out, const uint32_t *__restrict__ ids,
                                                   uint32_t n, uint32_t kstride) {
  constexpr int BUF = 512 + 2 * NH * 64; // doubles per wave per stage
  __shared__ __attribute__((aligned(16))) double lds[D * NW * BUF];
  const uint32_t wpx = gridDim.x >> 3, x = blockIdx.x & 7u, local = blockIdx.x >> 3;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const uint32_t groups = n / 8 / NW; // per XCD
  const uint32_t iters = (groups - local + wpx - 1) / wpx;
  constexpr int L = 4 + NH; // loads per brick
  constexpr int BEHIND = (D - 1) * (8 + L);
  const uint32_t v16 = lane * 16;
  auto issue = [&](uint32_t it) {
    const uint32_t g = local + (it < iters ? it : iters - 1) * wpx;
    const uint32_t b = __builtin_amdgcn_readfirstlane(ids[x * (n / 8) + g * NW + w]);
    gbytes self = (gbytes)(in + (size_t)b * 512);
    gbytes below = (gbytes)(in + (size_t)(b - kstride) * 512 + (8 - NH) * 64);
    gbytes above = (gbytes)(in + (size_t)(b + kstride) * 512);
    const uint32_t base = (uint32_t)(uintptr_t)(lds + ((it % D) * NW + w) * BUF);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16(base + q * 1024, self + q * 1024, v16);
#pragma unroll
    for (int q = 0; q < NH / 2; ++q) {
      dma16(base + 4096 + q * 1024, below + q * 1024, v16);
      dma16(base + 4096 + NH * 512 + q * 1024, above + q * 1024, v16);
    }
  };
#pragma unroll
  for (int q = 0; q < D - 1; ++q) issue(q);
  for (uint32_t it = 0; it < iters; ++it) {
    issue(it + D - 1); // past the end: the last brick again (keeps the counts of the waits below exact)
    // everything older than the D-1 younger bricks' loads and (once the pipeline is full) stores
    if (it < D - 1) wait_vm<(D - 1) * L>();
    else wait_vm<BEHIND>();
    const double *my = lds + ((it % D) * NW + w) * BUF;
    const uint32_t g = local + it * wpx;
    const uint32_t b = __builtin_amdgcn_readfirstlane(ids[x * (n / 8) + g * NW + w]);
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = my[k * 64 + lane];
#pragma unroll
    for (int p = 0; p < 2 * NH; ++p) r[p & 7] += my[512 + p * 64 + lane];
    double *dst = out + (size_t)b * 512;
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(r[k] * 1.0001, &dst[k * 64 + lane]);
  }
  wait_vm<0>();
}

template <int NH, int D, int NW>
static float run_pipe(const double *in, double *out, const uint32_t *ids, uint32_t n, uint32_t kstride, int iters, int R) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto go = [&]() { hipLaunchKernelGGL((k_pipe<NH, D, NW>), dim3(256 * R), dim3(64 * NW), 0, 0, in, out, ids, n, kstride); };
  for (int q = 0; q < 5; ++q) go();
  CK(hipEventRecord(e0));
  for (int q = 0; q < iters; ++q) go();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int OWN, int HALO, int NH>
static float run(const double *in, double *out, const uint32_t *ids, uint32_t n, uint32_t kstride, int iters, int nw, size_t extra) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto go = [&]() {
    if (nw == 8) {
      const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
      hipLaunchKernelGGL((k_probe<OWN, HALO, NH, 8>), dim3(wgs), dim3(512), extra, 0, in, out, ids, n, kstride);
    } else {
      const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
      hipLaunchKernelGGL((k_probe<OWN, HALO, NH, 4>), dim3(wgs), dim3(256), extra, 0, in, out, ids, n, kstride);
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
  return ms / iters;
}

int main(int argc, char **argv) {
  int N = 512, iters = 40, nw = 4, pipe = 0;
  size_t extra = 0;
  for (int a = 1; a < argc; ++a) {
    std::string o = argv[a];
    if (o == "-n") N = atoi(argv[++a]);
    else if (o == "-i") iters = atoi(argv[++a]);
    else if (o == "-w") nw = atoi(argv[++a]);
    else if (o == "-l") extra = (size_t)atol(argv[++a]);
    else if (o == "-p") pipe = 1;
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
  printf("# l1probe N=%d bricks=%u waves/wg=%d extra LDS=%zu B, %d launches each; GB/s = 16 B x cells / time (algorithmic)\n", N, n, nw, extra, iters);
  if (pipe) {
#define PROW(NHv, Dv, NWv, Rv)                                                                                                   \
  for (int rep = 0; rep < 2; ++rep) {                                                                                            \
    const float ms = run_pipe<NHv, Dv, NWv>(in, out, ids_d, n, ks, iters, Rv);                                                   \
    printf("pipe: planes/side %d depth %d waves/wg %d wgs/CU %d (%2d waves/CU, LDS %5.1f KiB/wg): %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", NHv, Dv, NWv, Rv, \
           NWv * Rv, Dv * NWv * (4.0 + NHv), ms, 16.0 * cells / ms / 1e6, 16.0 * cells / ms / 1e6 / 8000.0);                       \
    fflush(stdout);                                                                                                              \
  }
    PROW(0, 2, 4, 1) PROW(0, 2, 4, 2) PROW(0, 2, 4, 3) PROW(0, 2, 4, 4) PROW(0, 3, 4, 1) PROW(0, 3, 4, 2) PROW(0, 3, 4, 3) PROW(0, 4, 4, 2)
    PROW(0, 2, 8, 1) PROW(0, 2, 8, 2) PROW(0, 3, 8, 1) PROW(0, 3, 8, 2) PROW(0, 4, 8, 1)
    PROW(2, 2, 4, 2) PROW(2, 2, 4, 3) PROW(2, 3, 4, 2) PROW(2, 2, 8, 1) PROW(2, 3, 8, 1)
    PROW(4, 2, 4, 2) PROW(4, 3, 4, 1) PROW(4, 3, 4, 2) PROW(4, 2, 8, 1) PROW(4, 2, 8, 2)
    return 0;
  }
  const char *fl[] = {"dwordx2", "dwordx4", "dwordx4 sc1", "dwordx4 nt", "dwordx4 sc0 sc1", "lds-dma x4"};
#define ROW(O, H, NHv)                                                                                                          \
  for (int rep = 0; rep < 2; ++rep) {                                                                                           \
    const float ms = run<O, H, NHv>(in, out, ids_d, n, ks, iters, nw, extra);                                                   \
    printf("own %-16s halo %-16s planes/side %d (%4.1f KB through L1 per brick): %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", fl[O], \
           NHv ? fl[H] : "-", NHv, 4.0 + NHv * 1.0, ms, 16.0 * cells / ms / 1e6, 16.0 * cells / ms / 1e6 / 8000.0);                \
    fflush(stdout);                                                                                                             \
  }
  ROW(0, 0, 0) ROW(1, 1, 0) ROW(2, 2, 0) ROW(3, 3, 0) ROW(4, 4, 0) ROW(5, 5, 0)
  ROW(0, 0, 2) ROW(1, 1, 2) ROW(2, 1, 2) ROW(1, 2, 2) ROW(2, 2, 2) ROW(5, 5, 2) ROW(5, 1, 2)
  ROW(0, 0, 4) ROW(1, 1, 4) ROW(2, 1, 4) ROW(1, 2, 4) ROW(2, 2, 4) ROW(3, 1, 4) ROW(5, 5, 4) ROW(5, 1, 4)
  ROW(0, 0, 8) ROW(1, 1, 8) ROW(2, 1, 8) ROW(1, 2, 8) ROW(2, 2, 8) ROW(5, 5, 8)
  return 0;
}
