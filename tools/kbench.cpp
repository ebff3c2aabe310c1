// kbench -- native micro-benchmark of libbrk.so kernel variants (development tool, not part of the product).
//   kbench [-n N] [-s stencil_id] [-v v1,v2,...] [-i iters] [-d] [-r]
//     -d  BrickDecomp ordering + range API over [1,sep1) with two storages (weak/ shape); default is the
//         single/ shape: lexicographic (N/8+2)^3 grid, interleaved storage (step 1024), list API over the interior
//     -r  shuffle the brick list (worst-case locality)
// Every variant is checked against variant 1 (generic kernel) with a device-side max|a-b|.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "brk.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)
#define BK(x)                                                      \
  do {                                                             \
    int rc = (x);                                                  \
    if (rc) {                                                      \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc, brk_last_error()); \
      exit(3);                                                     \
    }                                                              \
  } while (0)

__global__ void k_fill(double *p, size_t n, unsigned long long seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = (i + seed) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0);
  }
}

// max |a-b| over the out fields of the listed bricks
__global__ void k_maxdiff(const double *a, const double *b, size_t step, const uint32_t *ids, uint32_t first, uint32_t n,
                          unsigned long long *res) {
  double m = 0;
  for (size_t w = blockIdx.x; w < n; w += gridDim.x) {
    uint32_t id = ids ? ids[w] : first + (uint32_t)w;
    if (id == 0xFFFFFFFFu) continue; // padding entry
    for (int e = threadIdx.x; e < 512; e += blockDim.x) {
      double d = fabs(a[(size_t)id * step + e] - b[(size_t)id * step + e]);
      if (!(d <= m)) m = d; // NaN propagates as "large"
    }
  }
  if (m != m) m = 1e300;
  atomicMax(res, (unsigned long long)__double_as_longlong(m));
}

// streaming ceilings with the same per-wave footprint as the stencil kernels (one 4 KiB brick per wave)
__global__ __launch_bounds__(256) void k_stream2(const double *in, double *out, size_t step, const uint32_t *ids, uint32_t n) {
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  uint32_t b = ids[w];
  int lane = threadIdx.x & 63;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = in[(size_t)b * step + k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 8; ++k) out[(size_t)b * step + k * 64 + lane] = v[k] * 1.0001;
}
__global__ __launch_bounds__(256) void k_stream4(const double *in, double *out, size_t step, const uint32_t *ids, uint32_t n) {
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  uint32_t b = ids[w];
  int lane = threadIdx.x & 63;
  const double2 *src = (const double2 *)(in + (size_t)b * step);
  double2 *dst = (double2 *)(out + (size_t)b * step);
  double2 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = src[k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k].x *= 1.0001;
    v[k].y *= 1.0001;
    dst[k * 64 + lane] = v[k];
  }
}

// linear float4 grid-stride copy (the guide's 6.29 TB/s reference shape)
__global__ __launch_bounds__(256) void k_copy_lin(const uint4 *in, uint4 *out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// one brick per wave, brick id = first + wave (no id list)
__global__ __launch_bounds__(256) void k_stream2_range(const double *in, double *out, size_t step, uint32_t first, uint32_t n) {
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  uint32_t b = first + w;
  int lane = threadIdx.x & 63;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = in[(size_t)b * step + k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 8; ++k) out[(size_t)b * step + k * 64 + lane] = v[k] * 1.0001;
}
// the same with non-temporal stores (what the stencil kernels use) / non-temporal loads as well
template <int NTL>
__global__ __launch_bounds__(256) void k_stream2_range_nt(const double *in, double *out, size_t step, uint32_t first, uint32_t n) {
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  uint32_t b = first + w;
  int lane = threadIdx.x & 63;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = NTL ? __builtin_nontemporal_load(&in[(size_t)b * step + k * 64 + lane]) : in[(size_t)b * step + k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k] * 1.0001, &out[(size_t)b * step + k * 64 + lane]);
}
// XCD-aware brick order as in the stencil kernels (workgroup -> XCD slab), 8 waves per workgroup, non-temporal stores
template <int NW, int XCD>
__global__ __launch_bounds__(64 * NW) void k_stream2_xcd_nt(const double *in, double *out, size_t step, uint32_t first, uint32_t n) {
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t wg = XCD ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
  uint32_t w = wg * NW + (threadIdx.x >> 6);
  if (w >= n) return;
  uint32_t b = first + w;
  int lane = threadIdx.x & 63;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = in[(size_t)b * step + k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k] * 1.0001, &out[(size_t)b * step + k * 64 + lane]);
}
// persistent waves: each wave walks bricks w, w+W, ... with the next brick's loads in flight while storing
__global__ __launch_bounds__(256) void k_stream2_persist(const double *in, double *out, size_t step, uint32_t first, uint32_t n) {
  const uint32_t W = gridDim.x * 4;
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (w >= n) return;
  double a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = in[(size_t)(first + w) * step + k * 64 + lane];
  for (; w < n; w += 2 * W) {
    uint32_t w1 = w + W < n ? w + W : w, w2 = w + 2 * W < n ? w + 2 * W : w;
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = in[(size_t)(first + w1) * step + k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(size_t)(first + w) * step + k * 64 + lane] = a[k] * 1.0001;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = in[(size_t)(first + w2) * step + k * 64 + lane];
    if (w + W < n) {
#pragma unroll
      for (int k = 0; k < 8; ++k) out[(size_t)(first + w1) * step + k * 64 + lane] = b[k] * 1.0001;
    }
  }
}

// what kind of interaction is the placement effect?  two read streams / two write streams over the same pair of storages
__global__ __launch_bounds__(256) void k_read2(const double *a, const double *b, double *c, uint32_t nb) {
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nb) return;
  int lane = threadIdx.x & 63;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[(size_t)w * 512 + k * 64 + lane] + b[(size_t)w * 512 + k * 64 + lane];
  if (s == 1.2345e300) c[w] = s; // never true: the loads stay
}
__global__ __launch_bounds__(256) void k_write2(double *a, double *b, uint32_t nb) {
  uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nb) return;
  int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    __builtin_nontemporal_store((double)lane, &a[(size_t)w * 512 + k * 64 + lane]);
    __builtin_nontemporal_store((double)k, &b[(size_t)w * 512 + k * 64 + lane]);
  }
}
// candidate placement probes (development): brick copy in the sweep's shape -- planes of PB bricks walked in order, XCD x (= blockIdx & 7)
// takes the x-th eighth of every plane, one wave per brick, NW waves per workgroup, non-temporal stores
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_probe_slab(const double *in, double *out, uint32_t nb, uint32_t PB) {
  const uint32_t per_xcd = gridDim.x >> 3, x = blockIdx.x & 7u, local = blockIdx.x >> 3;
  const uint32_t s = local * NW + (threadIdx.x >> 6); // position inside this XCD's share
  const uint32_t share = PB / 8, plane = s / share, off = s % share;
  const uint32_t b = plane * PB + x * share + off;
  (void)per_xcd;
  if (b >= nb) return;
  const int lane = threadIdx.x & 63;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = in[(size_t)b * 512 + k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k] * 1.0001, &out[(size_t)b * 512 + k * 64 + lane]);
}

int main(int argc, char **argv) {
  int N = 512, sid = BRK_7PT_COEFF, iters = 30, order = 0, kstack = 1, rounds = 2;
  bool decomp = false, shuffle = false, separate = false, stream = false, boxmode = false, reg = false, kpair = false, jpair = false, zigzag = false;
  long run = 4, nlev = 2; // -P / -J: entries per run; -P: k-levels per group (-K)
  long ichunk = 0;        // -C: with -o, a block of rows is cut along i into chunks of this many bricks, each swept over all k before the next
  bool one_alloc = false;
  int trials = 0;
  bool trial_copy = false, placed = false, declare_kpaired = false;
  unsigned trial_flags = 0;
  size_t out_shift = 0;   // -A: the output storage starts this many bytes into its allocation (DRAM channel / bank alignment of out against in)
  std::vector<int> variants = {1, 2, 5};
  for (int a = 1; a < argc; ++a) {
    std::string o = argv[a];
    if (o == "-n") N = atoi(argv[++a]);
    else if (o == "-s") sid = atoi(argv[++a]);
    else if (o == "-i") iters = atoi(argv[++a]);
    else if (o == "-r2") rounds = atoi(argv[++a]); // interleaved rounds over the variant list
    else if (o == "-d") decomp = true;
    else if (o == "-r") shuffle = true;
    else if (o == "-b") boxmode = true;     // brk_stencil_apply_box over the interior (lexi layouts only)
    else if (o == "-S") separate = true;   // lexicographic list but two storages (step 512)
    else if (o == "-o") order = atoi(argv[++a]); // 0 lexi; J>0: per-XCD j-slabs of 8 rows, swept in sub-blocks of J rows over all k
    else if (o == "-k") kstack = atoi(argv[++a]); // with -o: workgroup = 8 i-bricks x kstack k-levels
    else if (o == "-P") kpair = true;      // with -o: k-paired order -- runs of 4 i-neighbours at k followed by the same run at k+1 (k_star_kp)
    else if (o == "-J") jpair = true;      // with -o: j-paired order -- runs of i-neighbours of row j followed by the same run of row j+1
    else if (o == "-G") run = atol(argv[++a]); // with -P / -J: entries per run (4 or 8)
    else if (o == "-C") ichunk = atol(argv[++a]); // with -o (plain and -P orders): i-chunks of this many bricks (a multiple of 8) -- the distance between the two uses of a k-plane shrinks without giving up rows per block
    else if (o == "-K") nlev = atol(argv[++a]); // with -P: k-levels per group (2 = pairs, 4 = the k-stacked order of k_star_km)
    else if (o == "-R") reg = true;        // brk_adjacency_register: the table is promised immutable
    else if (o == "-Z") zigzag = true;     // list form: every other launch walks each XCD's share of the list backwards (groups of 8 kept)
    else if (o == "-A") out_shift = (size_t)atol(argv[++a]);
    else if (o == "-PP") kpair = declare_kpaired = true; // -P and the list is DECLARED k-paired (brk_stencil_apply_list_kpaired: the march kernels of the default build)
    else if (o == "-Q") placed = true; // storages from brk_storage_alloc: the output storage is placed against the input storage
    else if (o == "-F") trial_flags = (unsigned)atoi(argv[++a]); // -T: the fresh storages come from hipExtMallocWithFlags (1 fine-grained, 3 uncached, 4 physically contiguous)
    else if (o == "-Tc") { trials = atoi(argv[++a]); trial_copy = true; } // the same with a plain brick copy (non-temporal stores) instead of the stencil
    else if (o == "-T") trials = atoi(argv[++a]); // placement trials: the variant list is timed over `trials` freshly allocated output storages (earlier ones stay allocated)
    else if (o == "-A1") { one_alloc = true; out_shift = (size_t)atol(argv[++a]); } // in and out carved out of ONE allocation: out = in + round_up(size, 2 MiB) + shift
    else if (o == "-c") stream = true;     // also run the streaming-ceiling kernels
    else if (o == "-D") {                  // diagnostic builds of libbrk.so: timing experiment of the 7-point box kernel
      typedef void (*diag_fn)(int);
      diag_fn dg = (diag_fn)dlsym(RTLD_DEFAULT, "brk_debug_diag");
      if (dg) dg(atoi(argv[++a]));
    } else if (o == "-L") {                  // unused dynamic LDS per launch: fewer workgroups resident per CU (occupancy experiments)
      typedef void (*pad_fn)(size_t);
      pad_fn pad = (pad_fn)dlsym(RTLD_DEFAULT, "brk_debug_pad_lds");
      if (pad) pad((size_t)atol(argv[++a]));
    }
    else if (o == "-v") {
      variants.clear();
      char *tok = strtok(argv[++a], ",");
      while (tok) {
        variants.push_back(atoi(tok));
        tok = strtok(nullptr, ",");
      }
    }
  }
  std::vector<uint32_t> adj, ids;
  uint32_t nbricks, first = 0, nlist;
  size_t step;
  if (decomp) {
    brk_decomp *d;
    uint32_t dims[3] = {(uint32_t)N, (uint32_t)N, (uint32_t)N};
    BK(brk_decomp_build(dims, 8, &d));
    nbricks = brk_decomp_nbricks(d);
    adj.assign(brk_decomp_adj(d), brk_decomp_adj(d) + (size_t)nbricks * 27);
    uint32_t sep[3];
    brk_decomp_sep_pos(d, sep);
    first = 1;
    nlist = sep[1] - 1;
    step = 512;
    brk_decomp_free(d);
  } else {
    long SB = N / 8 + 2, dims[3] = {SB, SB, SB};
    nbricks = (uint32_t)(SB * SB * SB);
    std::vector<uint32_t> grid(nbricks);
    adj.resize((size_t)nbricks * 27);
    BK(brk_init_grid(dims, grid.data(), adj.data()));
    const long NB = SB - 2;
    if (order == 0) {
      for (long k = 1; k < SB - 1; ++k)
        for (long j = 1; j < SB - 1; ++j)
          for (long i = 1; i < SB - 1; ++i) ids.push_back(grid[(k * SB + j) * SB + i]);
    } else {
      // XCD x gets rows [x*NB/8, (x+1)*NB/8); inside: blocks of `order` rows swept over all k (k-reuse distance =
      // order*NB bricks), i fastest
      const long rows = NB / 8;
      const long ic = ichunk > 0 ? ichunk : NB;
      if (kpair) {
        for (long x = 0; x < 8; ++x)
          for (long jb = 0; jb < rows; jb += order)
            for (long i0 = 1; i0 < SB - 1; i0 += ic)
              for (long k = 1; k < SB - 1; k += nlev)
                for (long j = x * rows + jb; j < x * rows + jb + order && j < (x + 1) * rows; ++j)
                  for (long ib = i0; ib < i0 + ic && ib < SB - 1; ib += run)
                    for (long s2 = 0; s2 < nlev; ++s2)
                      for (long i = ib; i < ib + run; ++i) ids.push_back(k + s2 < SB - 1 && i < SB - 1 ? grid[((k + s2) * SB + (j + 1)) * SB + i] : BRK_LIST_SKIP);
      } else if (jpair) {
        for (long x = 0; x < 8; ++x)
          for (long jb = 0; jb < rows; jb += order)
            for (long k = 1; k < SB - 1; ++k)
              for (long j = x * rows + jb; j < x * rows + jb + order && j < (x + 1) * rows; j += 2)
                for (long ib = 1; ib < SB - 1; ib += run)
                  for (long s2 = 0; s2 < 2; ++s2)
                    for (long i = ib; i < ib + run; ++i)
                      ids.push_back(j + s2 < (x + 1) * rows && i < SB - 1 ? grid[(k * SB + (j + s2 + 1)) * SB + i] : BRK_LIST_SKIP);
      } else
      for (long x = 0; x < 8; ++x)
        for (long jb = 0; jb < rows; jb += order)
          for (long i0 = 1; i0 < SB - 1; i0 += ic)
            for (long k = 1; k < SB - 1; k += kstack)
              for (long j = x * rows + jb; j < x * rows + jb + order && j < (x + 1) * rows; ++j)
                for (long ib = i0; ib < i0 + ic && ib < SB - 1; ib += 8)
                  for (long s2 = 0; s2 < kstack; ++s2)
                    for (long i = ib; i < ib + 8; ++i) ids.push_back(grid[((k + s2) * SB + (j + 1)) * SB + i]);
    }
    nlist = (uint32_t)ids.size();
    step = separate ? 512 : 1024;
  }
  if (shuffle) {
    if (ids.empty())
      for (uint32_t q = 0; q < nlist; ++q) ids.push_back(first + q);
    std::mt19937 g(7);
    std::shuffle(ids.begin(), ids.end(), g);
  }
  const size_t elems = (size_t)nbricks * step;
  double *st_in, *st_out, *st_ref;
  uint32_t *adj_d, *ids_d = nullptr;
  unsigned long long *res_d;
  const size_t field_2m = (elems * 8 + (2u << 20) - 1) / (2u << 20) * (2u << 20);
  if (one_alloc && (decomp || separate)) {
    char *raw;
    CK(hipMalloc(&raw, 2 * field_2m + out_shift));
    st_in = (double *)raw;
    st_out = (double *)(raw + field_2m + out_shift);
    CK(hipMalloc(&st_ref, elems * 8));
    printf("# one allocation %p: out - in = %zu B\n", (void *)raw, field_2m + out_shift);
  } else if ((decomp || separate) && placed) {
    void *a = nullptr, *b = nullptr;
    brk_placement rep;
    BK(brk_storage_alloc(elems * 8, &a, nullptr));
    BK(brk_storage_alloc(elems * 8, &b, &rep));
    st_in = (double *)a, st_out = (double *)b;
    CK(hipMalloc(&st_ref, elems * 8));
    printf("# placed: %d candidates, chosen %d, probe sweep %.4f ms (slowest candidate %.4f)\n", rep.candidates, rep.chosen, rep.ms_chosen, rep.ms_worst);
  } else if (decomp || separate) {
    CK(hipMalloc(&st_in, elems * 8));
    char *raw;
    CK(hipMalloc(&raw, elems * 8 + out_shift));
    st_out = (double *)(raw + out_shift);
    CK(hipMalloc(&st_ref, elems * 8));
    printf("# in %p out %p: out - in = %td B\n", (void *)st_in, (void *)st_out, (char *)st_out - (char *)st_in);
  } else {
    CK(hipMalloc(&st_in, elems * 8));
    st_out = st_in + 512; // interleaved second field (stencils/3axis.cpp:25-28)
    CK(hipMalloc(&st_ref, elems * 8));
  }
  CK(hipMalloc(&adj_d, adj.size() * 4));
  CK(hipMalloc(&res_d, 8));
  CK(hipMemcpy(adj_d, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
  if (reg) BK(brk_adjacency_register(adj_d, nbricks));
  uint32_t *rids_d = nullptr;
  if (!ids.empty()) {
    CK(hipMalloc(&ids_d, ids.size() * 4));
    CK(hipMemcpy(ids_d, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    if (zigzag && ids.size() % 64 == 0) { // the same bricks, each XCD's eighth walked from its end (what was touched last is touched first)
      std::vector<uint32_t> r(ids.size());
      const size_t per = ids.size() / 8, groups = per / 8;
      for (size_t x = 0; x < 8; ++x)
        for (size_t g = 0; g < groups; ++g)
          for (size_t e = 0; e < 8; ++e) r[x * per + g * 8 + e] = ids[x * per + (groups - 1 - g) * 8 + e];
      CK(hipMalloc(&rids_d, r.size() * 4));
      CK(hipMemcpy(rids_d, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    }
  }
  unsigned long launches = 0;
  k_fill<<<4096, 256>>>(st_in, elems, 1);
  CK(hipMemset(st_ref, 0, elems * 8));
  CK(hipDeviceSynchronize());
  double coeff[129];
  for (int i = 0; i < 129; ++i) coeff[i] = 0.01 * (i + 1);
  const double *cp = (sid == BRK_7PT_COEFF || sid == BRK_27PT_COEFF) ? coeff : nullptr;
  double *ref_out = (decomp || separate) ? st_ref : st_ref + 512;

  auto launch = [&](double *out) {
    if (boxmode && !decomp) {
      uint32_t SBu = (uint32_t)(N / 8 + 2), tot[3] = {SBu, SBu, SBu}, lo3[3] = {1, 1, 1}, cnt[3] = {SBu - 2, SBu - 2, SBu - 2};
      BK(brk_stencil_apply_box(sid, adj_d, st_in, out, step, 0, tot, lo3, cnt, cp, nullptr));
    } else if (ids_d && kpair && declare_kpaired)
      BK(brk_stencil_apply_list_kpaired(sid, adj_d, st_in, out, step, ids_d, nlist, cp, nullptr));
    else if (ids_d)
      BK(brk_stencil_apply_list(sid, adj_d, st_in, out, step, (rids_d && (launches++ & 1)) ? rids_d : ids_d, nlist, cp, nullptr));
    else
      BK(brk_stencil_apply_range(sid, adj_d, st_in, out, step, first, first + nlist, cp, nullptr));
  };
  // reference output with the generic kernel (variant 1) where the library was built with VARIANTS=1, else with the default kernel
  typedef int (*variant_fn)(int, int);
  variant_fn set_variant = (variant_fn)dlsym(RTLD_DEFAULT, "brk_set_variant"); // development builds of libbrk.so only (make DEV=1 / VARIANTS=1)
  auto brk_set_variant = [&](int id, int v) { return set_variant ? set_variant(id, v) : (v == 0 ? (int)BRK_OK : (int)BRK_EINVAL); };
  const bool have_variants = brk_set_variant(sid, 1) == BRK_OK;
  if (!have_variants) BK(brk_set_variant(sid, 0));
  launch(ref_out);
  CK(hipDeviceSynchronize());

  printf("# N=%d stencil=%d bricks=%u list=%u %s%s order=%d step=%zu iters=%d\n", N, sid, nbricks, nlist,
         decomp ? "decomp-range" : "lexi-list", shuffle ? " shuffled" : "", order, step, iters);
  const double cells = (double)N * N * N;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (stream && ids_d) {
    const char *names[] = {"stream2(ids)", "stream4(ids)", "copy_lin float4 2048 blocks", "copy_lin float4 8192 blocks",
                           "stream2_range", "stream2_persist 2048 blocks", "stream2_persist 4096 blocks",
                           "stream2_range nt stores", "stream2_range nt loads+stores", "stream2 xcd order 8 waves nt stores", "stream2 xcd order 4 waves nt stores",
                           "stream2 natural order 8 waves nt stores", "stream2 xcd order 2 waves nt stores"};
    for (int which = 0; which < 13; ++which) {
      auto go = [&]() {
        if (which == 0) k_stream2<<<(nlist + 3) / 4, 256>>>(st_in, st_out, step, ids_d, nlist);
        else if (which == 1) k_stream4<<<(nlist + 3) / 4, 256>>>(st_in, st_out, step, ids_d, nlist);
        else if (which == 2) k_copy_lin<<<2048, 256>>>((const uint4 *)st_in, (uint4 *)st_ref, (size_t)nlist * 512 / 2);
        else if (which == 3) k_copy_lin<<<8192, 256>>>((const uint4 *)st_in, (uint4 *)st_ref, (size_t)nlist * 512 / 2);
        else if (which == 4) k_stream2_range<<<(nlist + 3) / 4, 256>>>(st_in, st_out, step, 0, nlist);
        else if (which == 5) k_stream2_persist<<<2048, 256>>>(st_in, st_out, step, 0, nlist);
        else if (which == 6) k_stream2_persist<<<4096, 256>>>(st_in, st_out, step, 0, nlist);
        else if (which == 7) k_stream2_range_nt<0><<<(nlist + 3) / 4, 256>>>(st_in, st_out, step, 0, nlist);
        else if (which == 8) k_stream2_range_nt<1><<<(nlist + 3) / 4, 256>>>(st_in, st_out, step, 0, nlist);
        else if (which == 9) k_stream2_xcd_nt<8, 1><<<(((nlist + 7) / 8 + 7) / 8) * 8, 512>>>(st_in, st_out, step, 0, nlist);
        else if (which == 10) k_stream2_xcd_nt<4, 1><<<(((nlist + 3) / 4 + 7) / 8) * 8, 256>>>(st_in, st_out, step, 0, nlist);
        else if (which == 11) k_stream2_xcd_nt<8, 0><<<(((nlist + 7) / 8 + 7) / 8) * 8, 512>>>(st_in, st_out, step, 0, nlist);
        else k_stream2_xcd_nt<2, 1><<<(((nlist + 1) / 2 + 7) / 8) * 8, 128>>>(st_in, st_out, step, 0, nlist);
      };
      for (int w = 0; w < 3; ++w) go();
      CK(hipEventRecord(e0));
      for (int it = 0; it < iters; ++it) go();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      printf("%-30s %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)\n", names[which], ms, 16.0 * cells / ms / 1e6,
             16.0 * cells / ms / 1e6 / 80.0);
    }
  }
  // variant codes >= 100 (A/B inside one process, interleaved rounds): 1xx = adjacency registered (brk_adjacency_register) for
  // that variant's launches, 2xx = diagnostic build, diag 1, 3xx = registered + diag 16
  typedef void (*diag_fn2)(int);
  diag_fn2 set_diag = (diag_fn2)dlsym(RTLD_DEFAULT, "brk_debug_diag");
  if (trials > 0 && (decomp || separate)) {
    // does the PLACEMENT of the two storages matter?  Same process, same kernel: `trials` freshly allocated storages (all kept), the
    // sweep timed from storage A into storage B for A = the original input and two of the fresh ones
    BK(brk_adjacency_register(adj_d, nbricks));
    std::vector<double *> buf = {st_in, st_out, st_ref};
    if (trial_flags & 8u) { // -F 12 / -F 8: ONE allocation (physically contiguous with bit 4) carved into `trials` storages, 1 GiB + 128 MiB apart
      const size_t pitch = ((size_t)1 << 30) + ((size_t)128 << 20);
      if (elems * 8 > pitch) { fprintf(stderr, "storage larger than the carving pitch\n"); return 2; }
      char *raw;
      if (trial_flags & 4u) CK(hipExtMallocWithFlags((void **)&raw, pitch * trials, 4u));
      else CK(hipMalloc(&raw, pitch * trials));
      buf.clear();
      for (int t = 0; t < trials; ++t) buf.push_back((double *)(raw + pitch * t));
      printf("# one allocation of %.1f GiB at %p%s, storages every %.3f GiB\n", pitch * trials / 1073741824.0, (void *)raw, (trial_flags & 4u) ? " (hipDeviceMallocContiguous)" : "", pitch / 1073741824.0);
    } else
    for (int t = 0; t < trials; ++t) {
      char *raw;
      if (trial_flags) CK(hipExtMallocWithFlags((void **)&raw, elems * 8, trial_flags));
      else CK(hipMalloc(&raw, elems * 8));
      buf.push_back((double *)raw);
    }
    auto timed = [&](const double *src, double *dst) {
      auto go = [&]() {
        uint32_t SBu = (uint32_t)(N / 8 + 2), tot[3] = {SBu, SBu, SBu}, lo3[3] = {1, 1, 1}, cnt[3] = {SBu - 2, SBu - 2, SBu - 2};
        if (trial_copy) k_stream2_range_nt<0><<<(nbricks + 3) / 4, 256>>>(src, dst, step, 0, nbricks);
        else if (boxmode) BK(brk_stencil_apply_box(sid, adj_d, src, dst, step, 0, tot, lo3, cnt, cp, nullptr));
        else BK(brk_stencil_apply_list(sid, adj_d, src, dst, step, ids_d, nlist, cp, nullptr));
      };
      for (int w = 0; w < 5; ++w) go();
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) go();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / iters);
      }
      return best;
    };
    const size_t srcs[3] = {0, buf.size() / 3, 2 * buf.size() / 3};
    for (size_t si : srcs) {
      k_fill<<<4096, 256>>>(buf[si], elems, 1);
      CK(hipDeviceSynchronize());
      for (size_t t = 0; t < buf.size(); ++t) {
        if (t == si) continue;
        const float ms = timed(buf[si], buf[t]);
        double probe_ms = 0.0; // the library's own placement measurement of the same pair (overwrites buf[t]: fine, it is an output)
        BK(brk_placement_probe(buf[si], buf[t], elems * 8, nullptr, &probe_ms));
        auto time_k = [&](auto go) {
          for (int w = 0; w < 2; ++w) go();
          CK(hipEventRecord(e0));
          for (int it = 0; it < 6; ++it) go();
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float m2;
          CK(hipEventElapsedTime(&m2, e0, e1));
          return m2 / 6;
        };
        const uint32_t nb = nbricks;
        const float c_rr = time_k([&]() { k_read2<<<(nb + 3) / 4, 256>>>(buf[si], buf[t], st_ref, nb); });
        const float c_ww = trial_copy ? time_k([&]() { k_write2<<<(nb + 3) / 4, 256>>>(buf[0 == si ? 1 : 0], buf[t], nb); }) : 0.f; // -Tc: also two write streams (destroys storage 0/1: run last)
        const float c_range = time_k([&]() { k_stream2_range_nt<0><<<(nb + 3) / 4, 256>>>(buf[si], buf[t], 512, 0, nb); });
        const float c_xcd8 = time_k([&]() { k_stream2_xcd_nt<8, 1><<<(((nb + 7) / 8 + 7) / 8) * 8, 512>>>(buf[si], buf[t], 512, 0, nb); });
        const float c_slab = time_k([&]() { k_probe_slab<8><<<(((nb + 7) / 8 + 7) / 8) * 8, 512>>>(buf[si], buf[t], nb, 4096); });
        const float c_slab2 = time_k([&]() { k_probe_slab<8><<<(((nb + 7) / 8 + 7) / 8) * 8, 512>>>(buf[si], buf[t], nb, 512); });
        printf("   two read streams over the pair %.4f ms%s\n", c_rr, trial_copy ? "" : "");
        if (trial_copy) printf("   two write streams (storage %d and storage %zu) %.4f ms\n", 0 == si ? 1 : 0, t, c_ww);
        printf("in = storage %2zu  out = storage %2zu  kernel %.4f ms | library probe (7-pt sweep over the probe grid) %.4f ms  brick copy range %.4f  xcd eighths %.4f  slab 16 MiB %.4f  slab 2 MiB %.4f\n", si, t, ms,
               probe_ms, c_range, c_xcd8, c_slab, c_slab2);
        fflush(stdout);
      }
    }
    (void)hipDeviceSynchronize();
    brk_adjacency_release(adj_d);
    return 0;
  }
  for (int round = 0; round < rounds; ++round)
    for (int vcode : variants) {
      const int v = vcode % 100;
      int mode = vcode / 100;
      const bool reg_diag = mode >= 1000; // 1000xx + 100 * bits: registered table AND development bits (DEV=1 builds: 256 = cube kernels load every adjacency row)
      if (reg_diag) {
        BK(brk_adjacency_register(adj_d, nbricks));
        if (set_diag) set_diag(mode - 1000);
        mode = 1;
      } else {
        if (mode == 1 || mode == 3) BK(brk_adjacency_register(adj_d, nbricks));
        if (set_diag && mode >= 2 && mode < 10) set_diag(mode == 2 ? 1 : 16);
        if (set_diag && mode >= 10) set_diag(mode - 10); // 10xx + 100 * bits: ablation bits of the cube kernels (diagnostic builds)
      }
      struct Undo {
        int mode; const uint32_t *adj; diag_fn2 sd; bool rd;
        ~Undo() { if (mode == 1 || mode == 3) { (void)hipDeviceSynchronize(); brk_adjacency_release(adj); } if (sd && (mode >= 2 || rd)) sd(0); }
      } undo{mode, adj_d, set_diag, reg_diag};
      if (brk_set_variant(sid, v) != BRK_OK) {
        printf("variant %d  not in this build of libbrk.so (make -C bricklib_amd/csrc VARIANTS=1)\n", vcode);
        continue;
      }
      for (int w = 0; w < 3; ++w) launch(st_out);
      CK(hipEventRecord(e0));
      for (int it = 0; it < iters; ++it) launch(st_out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      CK(hipMemset(res_d, 0, 8));
      k_maxdiff<<<1024, 256>>>(st_out, ref_out, step, ids_d, first, nlist, res_d);
      unsigned long long bits;
      CK(hipMemcpy(&bits, res_d, 8, hipMemcpyDeviceToHost));
      double md;
      memcpy(&md, &bits, 8);
      printf("variant %d  %.4f ms  %.1f GStencil/s  %.0f GB/s (%.1f%% of 8 TB/s)  maxdiff_vs_v1 %.3g\n", vcode, ms,
             cells / ms / 1e6, 16.0 * cells / ms / 1e6, 16.0 * cells / ms / 1e6 / 80.0, md);
      fflush(stdout);
      // diagnostic builds of libbrk.so (make -C bricklib_amd/csrc EXTRA_HIPFLAGS=-DBRK_STAMP) export per-phase shader-clock
      // totals of k_cube_dense (in-kernel s_memtime stamps, one row per list slot)
      typedef int (*stamp_fn)(unsigned long long *, int);
      static stamp_fn stamps = (stamp_fn)dlsym(RTLD_DEFAULT, "brk_debug_stamps");
      typedef int (*stampw_fn)(unsigned long long (*)[8]);
      static stampw_fn stamps_w = (stampw_fn)dlsym(RTLD_DEFAULT, "brk_debug_stamps_by_wave");
      if (stamps_w) {
        unsigned long long tw[4][8];
        if (stamps_w(tw) == 0)
          for (int w4 = 0; w4 < 4; ++w4)
            if (tw[w4][7])
              printf("   wave %d of its group: ids %.0f | adjacency %.0f | issue %.0f | loads back %.0f | stage+barriers %.0f | compute+store %.0f\n", w4,
                     (double)tw[w4][0] / tw[w4][7], (double)tw[w4][1] / tw[w4][7], (double)tw[w4][2] / tw[w4][7], (double)tw[w4][3] / tw[w4][7],
                     (double)tw[w4][4] / tw[w4][7], (double)tw[w4][5] / tw[w4][7]);
      }
      if (stamps) {
        unsigned long long t[8];
        if (stamps(t, 1) == 0 && t[7]) {
          const double w = (double)t[7];
          printf("   stamps per wave (clk): ids %.0f | adjacency %.0f | issue faces+gathers %.0f | loads back %.0f | stage+barriers %.0f | compute+store %.0f  (%llu waves)\n",
                 t[0] / w, t[1] / w, t[2] / w, t[3] / w, t[4] / w, t[5] / w, t[7]);
        }
      }
    }
  return 0;
}
