// examples/strong_hip.cpp -- the reference's strong-scaling driver shape (strong/main.cu:85-510, strong/args.cpp) on the MI355X
// engine, C++ only, through the C ABI:
//   global d^3 periodic domain cut into (d/s)^3 subdomains in Z-Morton order, contiguous id ranges per rank (brk_strong_partition,
//   brk_strong_neighbor = getrank); all subdomains of a rank in ONE storage with a block-diagonal adjacency table, so that a sweep
//   over every subdomain is one launch (the reference launches a 3-D grid over a table of Brick3D, main.cu:85-99,464-476);
//   ghost runs whose source subdomain lives on the same rank are ALIASED: every adjacency entry that names such a ghost brick is
//   pointed at the source subdomain's own brick, so these runs are neither copied nor swept (same subdomain cells bit for bit; the
//   adjacency-table form of the reference's memory-mapped ghost zones, strong/mmap.cu) -- or, with a 4th argument `copy`, device
//   links as in the reference (RegionLink, main.cu:201-208); the other runs are packed
//   into one buffer per peer rank: ONE send + ONE recv per peer (brk_xplan_build_tables -> brk_exchange_begin / transport / end).
// One process plays `size` ranks on one GPU (the transport between them is brk_exchange_loopback -- device copies in message
// order; with MPI each rank keeps only its own entry and calls brk_exchange_transport on its ncclComm_t instead), runs one
// brick_func (exchange + ST_ITER sweeps of the 13-point stencil, main.cpp:359-365) and validates every subdomain against the
// global array swept on the CPU -- the reference's `-v` option.
//   build: make -C examples strong_hip        run: examples/strong_hip [d=64] [s=16] [ranks=3] [alias|copy]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "brick-hip.h"
#include "brick-mpi.h"
#include "bricktime.h"
#include "brk.h"
#include "crash_trace.h"

#define BK(x)                                            \
  do {                                                   \
    if ((x) != BRK_OK) {                                 \
      fprintf(stderr, "%s: %s\n", #x, brk_last_error()); \
      exit(1);                                           \
    }                                                    \
  } while (0)

int MPI_ITER = 20;
typedef BrickDecomp<3, 8, 8, 8> Decomp;

// input field: a hash of the GLOBAL cell coordinate (the reference fills with random numbers and compares with its own array sweep)
static double field(long x, long y, long z, long d) {
  unsigned long long h = 42ull + (unsigned long long)((z * d + y) * d + x) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 30, h *= 0xBF58476D1CE4E5B9ull, h ^= h >> 27, h *= 0x94D049BB133111EBull, h ^= h >> 31;
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

struct Rank {
  int rank = 0;
  uint64_t l = 0, r = 0; // Z-Morton ids [l, r) of the subdomains it owns (strong/args.cpp:104-113)
  size_t nsub = 0;
  double *st[3] = {nullptr, nullptr, nullptr}; // three storages of nsub * nb bricks (main.cpp:359-365 rotates them)
  uint32_t *adj = nullptr, *list_all = nullptr, *list_domain = nullptr;
  uint32_t n_all = 0, n_domain = 0;
  brk_xplan *plan = nullptr;
  brk_exchange_view *view = nullptr;
};

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0); // the report survives whatever happens later, also into a pipe
  brk_crash::install(); // a crash prints its thread and call stack before the process dies by the signal (crash_trace.h)
  brk_crash::census("first line of main"); // BRK_CENSUS=1: threads born in library constructors are the ones listed here
  const long d = argc > 1 ? atol(argv[1]) : 64, s = argc > 2 ? atol(argv[2]) : 16;
  const int size = argc > 3 ? atoi(argv[3]) : 3;
  const bool alias_local = !(argc > 4 && std::string(argv[4]) == "copy");
  const uint64_t subdim = (uint64_t)(d / s), allsubs = subdim * subdim * subdim;
  if (d % s || s % 8 || (subdim & (subdim - 1)) || size < 1 || (uint64_t)size > allsubs) {
    fprintf(stderr, "need d %% s == 0, s %% 8 == 0, a power of two of subdomains per axis and at most one rank per subdomain\n");
    return 1;
  }
  // ONE decomposition shared by all subdomains (main.cpp:91-94)
  Decomp bd({(unsigned)s, (unsigned)s, (unsigned)s}, 8);
  bd.initialize(skin3d_good);
  const uint32_t nb = bd.nbricks();
  BrickInfo<3> one = bd.getBrickInfo();
  const size_t esz = sizeof(bElem), bbytes = 512 * esz;
  const long SB = s / 8 + 2;

  std::vector<Rank> ranks(size);
  for (int r = 0; r < size; ++r) {
    Rank &R = ranks[r];
    R.rank = r;
    uint64_t lr[2];
    brk_strong_partition(allsubs, size, r, lr);
    R.l = lr[0], R.r = lr[1], R.nsub = (size_t)(lr[1] - lr[0]);
    const size_t elems = R.nsub * nb * 512;
    for (int w = 0; w < 3; ++w) {
      void *placed = nullptr; // placed in HBM against the storages allocated before it (brk.h: brk_storage_alloc)
      BK(brk_storage_alloc(elems * esz, &placed, nullptr));
      R.st[w] = static_cast<double *>(placed);
      gpuCheck(hipMemset(R.st[w], 0, elems * esz));
    }
    // fill: subdomain with Z-Morton id z sits at block coordinates (zmort(z,0), zmort(z,1), zmort(z,2)) * s
    std::vector<double> host(elems, 0.0);
    for (size_t m = 0; m < R.nsub; ++m) {
      const uint64_t z = R.l + m;
      const long ox = (long)brk_zmort_get(z, 0) * s, oy = (long)brk_zmort_get(z, 1) * s, oz = (long)brk_zmort_get(z, 2) * s;
      for (long kb = 1; kb < SB - 1; ++kb)
        for (long jb = 1; jb < SB - 1; ++jb)
          for (long ib = 1; ib < SB - 1; ++ib) {
            double *b = &host[(m * nb + bd[(int)kb][(int)jb][(int)ib]) * 512];
            for (long k = 0; k < 8; ++k)
              for (long j = 0; j < 8; ++j)
                for (long i = 0; i < 8; ++i)
                  b[(k * 8 + j) * 8 + i] = field(ox + (ib - 1) * 8 + i, oy + (jb - 1) * 8 + j, oz + (kb - 1) * 8 + k, d);
          }
    }
    gpuCheck(hipMemcpy(R.st[0], host.data(), elems * esz, hipMemcpyHostToDevice));

    // ---- exchange plan (main.cu:188-395 at run granularity); both ends enumerate a peer's runs in the same canonical order
    // (receiver Z-Morton id, run index), so the buffers need no header
    auto off = [&](size_t m, uint32_t pos) { return (uint64_t)((m * nb + pos) * bbytes); };
    std::vector<brk_xcopy> copies;
    std::vector<uint32_t> alias(R.nsub * nb); // brick -> the brick that holds its data
    for (size_t b = 0; b < alias.size(); ++b) alias[b] = (uint32_t)b;
    typedef std::tuple<uint64_t, uint32_t, size_t> Key; // (receiver zmort, run, local subdomain)
    std::map<int, std::vector<Key>> recv, send;
    for (size_t m = 0; m < R.nsub; ++m) {
      const uint64_t z = R.l + m;
      for (uint32_t q = 0; q < (uint32_t)bd.ghost.size(); ++q) {
        int owner = 0, slot = 0;
        // receive side: ghost run q of z comes from the subdomain in direction ghost[q].neighbor
        brk_strong_neighbor(bd.ghost[q].neighbor.set, z, subdim, allsubs, size, &owner, &slot);
        if (owner == r && alias_local)
          for (uint32_t t = 0; t < bd.ghost[q].len; ++t) alias[m * nb + bd.ghost[q].pos + t] = (uint32_t)((size_t)slot * nb + bd.skin[q].pos + t);
        else if (owner == r)
          copies.push_back(brk_xcopy{0, 0, BRK_SP_STORAGE, BRK_SP_STORAGE, -1, -1, off((size_t)slot, bd.skin[q].pos), off(m, bd.ghost[q].pos),
                                     (uint64_t)bd.ghost[q].len * bbytes});
        else
          recv[owner].emplace_back(z, q, m);
        // send side: skin run q of z goes to the subdomain in direction skin[q].neighbor
        const uint64_t zr = brk_strong_neighbor(bd.skin[q].neighbor.set, z, subdim, allsubs, size, &owner, &slot);
        if (owner != r) send[owner].emplace_back(zr, q, m);
      }
    }
    std::vector<int> peers;
    for (auto &kv : recv) peers.push_back(kv.first);
    for (auto &kv : send) peers.push_back(kv.first);
    std::sort(peers.begin(), peers.end());
    peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
    std::vector<brk_xmsg> msgs;
    for (int p : peers) {
      uint64_t so = 0, ro = 0;
      auto &sv = send[p], &rv = recv[p];
      std::sort(sv.begin(), sv.end());
      std::sort(rv.begin(), rv.end());
      for (auto &k : sv) {
        const uint32_t q = std::get<1>(k);
        const uint64_t n = (uint64_t)bd.skin[q].len * bbytes;
        copies.push_back(brk_xcopy{0, 0, BRK_SP_STORAGE, BRK_SP_SEND, -1, p, off(std::get<2>(k), bd.skin[q].pos), so, n});
        so += n;
      }
      for (auto &k : rv) {
        const uint32_t q = std::get<1>(k);
        const uint64_t n = (uint64_t)bd.ghost[q].len * bbytes;
        copies.push_back(brk_xcopy{0, 1, BRK_SP_RECV, BRK_SP_STORAGE, p, -1, ro, off(std::get<2>(k), bd.ghost[q].pos), n});
        ro += n;
      }
      if (ro) msgs.push_back(brk_xmsg{0, 0, p, BRK_SP_RECV, 0, ro});
      if (so) msgs.push_back(brk_xmsg{0, 1, p, BRK_SP_SEND, 0, so});
    }
    // block-diagonal adjacency: brick b of subdomain m is brick m * nb + b of the rank's storage; aliased ghost bricks are replaced
    // by their sources and left out of the sweep over subdomains + ghost rings
    std::vector<uint32_t> adj((size_t)R.nsub * nb * 27), all, dom;
    for (size_t m = 0; m < R.nsub; ++m) {
      for (size_t q = 0; q < (size_t)nb * 27; ++q) adj[m * nb * 27 + q] = alias[(&one.adj[0][0])[q] + (uint32_t)(m * nb)];
      for (uint32_t b = 1; b < bd.sep_pos[2]; ++b)
        if (alias[m * nb + b] == (uint32_t)(m * nb) + b) all.push_back((uint32_t)(m * nb) + b);    // domain + ghost bricks
      for (uint32_t b = 1; b < bd.sep_pos[1]; ++b) dom.push_back((uint32_t)(m * nb) + b);    // the subdomain itself
    }
    R.n_all = (uint32_t)all.size(), R.n_domain = (uint32_t)dom.size();
    gpuCheck(hipMalloc(&R.adj, adj.size() * 4));
    gpuCheck(hipMemcpy(R.adj, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
    gpuCheck(hipMalloc(&R.list_all, all.size() * 4));
    gpuCheck(hipMemcpy(R.list_all, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    gpuCheck(hipMalloc(&R.list_domain, dom.size() * 4));
    gpuCheck(hipMemcpy(R.list_domain, dom.data(), dom.size() * 4, hipMemcpyHostToDevice));
    BK(brk_xplan_build_tables(r, (uint32_t)copies.size(), copies.data(), (uint32_t)msgs.size(), msgs.data(), &R.plan));
    BK(brk_exchange_view_create(R.plan, R.st[0], &R.view));
    printf("rank %d: subdomains [%llu, %llu), %zu peers, %.2f MB leave the rank per exchange, %.2f MB are copied inside it (links, pack, unpack), "
           "%u of %u bricks swept per sweep (%s)\n", r,
           (unsigned long long)R.l, (unsigned long long)R.r, peers.size(), brk_xplan_bytes_remote(R.plan) / 1e6,
           brk_xplan_bytes_local(R.plan) / 1e6, R.n_all, (uint32_t)(R.nsub * (bd.sep_pos[2] - 1)), alias_local ? "local ghosts aliased" : "local ghosts copied");
  }

  brk_crash::census("after the first HIP calls (storages, tables, views of every rank)");
  brk_crash::snapshot(); // libraries the runtime loads lazily are mapped now
  std::vector<brk_exchange_view *> views;
  for (auto &R : ranks) views.push_back(R.view);
  const int S = brk_stencil_iters(BRK_MPI13);
  // brick_func of strong/main.cu:397-510: same-rank links + pack, transport, unpack, then ST_ITER sweeps
  auto brick_func = [&]() {
    for (auto &R : ranks) BK(brk_exchange_begin(R.view, 0, nullptr));
    BK(brk_exchange_loopback(views.data(), (uint32_t)views.size(), 0, nullptr));
    for (auto &R : ranks) BK(brk_exchange_end(R.view, 0, nullptr));
    for (auto &R : ranks) {
      // 1 <- 0; (2 <- 1; 1 <- 2) x (S/2 - 1); 0 <- 1 over the subdomains only (main.cpp:359-365)
      BK(brk_stencil_apply_list(BRK_MPI13, R.adj, R.st[0], R.st[1], 512, R.list_all, R.n_all, nullptr, nullptr));
      for (int t = 0; t < S / 2 - 1; ++t) {
        BK(brk_stencil_apply_list(BRK_MPI13, R.adj, R.st[1], R.st[2], 512, R.list_all, R.n_all, nullptr, nullptr));
        BK(brk_stencil_apply_list(BRK_MPI13, R.adj, R.st[2], R.st[1], 512, R.list_all, R.n_all, nullptr, nullptr));
      }
      BK(brk_stencil_apply_list(BRK_MPI13, R.adj, R.st[1], R.st[0], 512, R.list_domain, R.n_domain, nullptr, nullptr));
    }
  };
  brick_func();
  gpuCheck(hipDeviceSynchronize());

  // ---- validation (-v of the reference): the global periodic array swept S times on the CPU (stencils/fake.h:73-79 weights)
  std::vector<double> ga((size_t)d * d * d), gb(ga.size());
  for (long z = 0; z < d; ++z)
    for (long y = 0; y < d; ++y)
      for (long x = 0; x < d; ++x) ga[((size_t)z * d + y) * d + x] = field(x, y, z, d);
  const double B0 = 0.4, B1 = 0.07, B2 = 0.03;
  auto at = [&](const std::vector<double> &g, long x, long y, long z) { return g[((size_t)((z + d) % d) * d + (y + d) % d) * d + (x + d) % d]; };
  for (int t = 0; t < S; ++t) {
#pragma omp parallel for
    for (long z = 0; z < d; ++z)
      for (long y = 0; y < d; ++y)
        for (long x = 0; x < d; ++x)
          gb[((size_t)z * d + y) * d + x] =
              B2 * (at(ga, x, y, z + 2) + at(ga, x, y, z - 2) + at(ga, x, y + 2, z) + at(ga, x, y - 2, z) + at(ga, x + 2, y, z) + at(ga, x - 2, y, z)) +
              B1 * (at(ga, x, y, z + 1) + at(ga, x, y, z - 1) + at(ga, x, y + 1, z) + at(ga, x, y - 1, z) + at(ga, x + 1, y, z) + at(ga, x - 1, y, z)) +
              B0 * at(ga, x, y, z);
    ga.swap(gb);
  }
  brk_crash::census("after the first omp parallel regions (CPU validation sweeps)");
  double worst = 0.0;
  for (auto &R : ranks) {
    std::vector<double> host(R.nsub * nb * 512);
    gpuCheck(hipMemcpy(host.data(), R.st[0], host.size() * esz, hipMemcpyDeviceToHost));
    for (size_t m = 0; m < R.nsub; ++m) {
      const uint64_t z = R.l + m;
      const long ox = (long)brk_zmort_get(z, 0) * s, oy = (long)brk_zmort_get(z, 1) * s, oz = (long)brk_zmort_get(z, 2) * s;
      for (long kb = 1; kb < SB - 1; ++kb)
        for (long jb = 1; jb < SB - 1; ++jb)
          for (long ib = 1; ib < SB - 1; ++ib) {
            const double *b = &host[(m * nb + bd[(int)kb][(int)jb][(int)ib]) * 512];
            for (long k = 0; k < 8; ++k)
              for (long j = 0; j < 8; ++j)
                for (long i = 0; i < 8; ++i)
                  worst = std::max(worst, std::fabs(b[(k * 8 + j) * 8 + i] - at(ga, ox + (ib - 1) * 8 + i, oy + (jb - 1) * 8 + j, oz + (kb - 1) * 8 + k)));
          }
    }
  }
  const bool ok = worst <= 1e-12;
  printf("%s: max |GPU - CPU| over all %llu subdomains = %.3e after %d sweeps\n", ok ? "Validated" : "INVALID", (unsigned long long)allsubs, worst, S);

  // timing (time_mpi protocol: warm-up, then MPI_ITER calls); data keeps evolving, which is what the reference's loop does too
  {
    int cnt = 0;
    const double sec = time_mpi([&]() { brick_func(); gpuCheck(hipDeviceSynchronize()); }, cnt, bd);
    printf("strong brick_func: %.3f ms per exchange + %d sweeps of %llu subdomains, %.1f GStencil/s (%d iterations)\n", sec * 1e3, S,
           (unsigned long long)allsubs, (double)d * d * d * S / sec / 1e9, cnt);
  }
  for (auto &R : ranks) {
    brk_exchange_view_free(R.view);
    brk_xplan_free(R.plan);
    for (int w = 0; w < 3; ++w) BK(brk_storage_free(R.st[w]));
    gpuCheck(hipFree(R.adj));
    gpuCheck(hipFree(R.list_all));
    gpuCheck(hipFree(R.list_domain));
  }
  free(one.adj);
  // orderly end (brick-gpu.h): OpenMP workers joined, the library's device-side caches released, the device idle -- on this thread,
  // before exit() runs the handlers of the HIP runtime beside its own threads
  brkFinalize();
  brk_crash::census("just before return");
  brk_crash::snapshot();
  return ok ? 0 : 1;
}
